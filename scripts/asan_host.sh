#!/bin/bash
# Host-side AddressSanitizer + UBSan pass (CPU only; GPU sanitizers are not available on this pool):
#   * the engine's host code (argument validation + HBM image construction, rk_db_validate) built with
#     -fsanitize=address,undefined -fno-gpu-sanitize, exercised by tests/test_cabi.py -k validate
#   * the C oracles built with the same sanitizers, exercised by tests/test_oracle_golden.py and tests/test_oracle_build.py
set -e
cd "$(dirname "$0")/.."
RT=$(/opt/rocm/lib/llvm/bin/clang --print-file-name=libclang_rt.asan-x86_64.so)
hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -shared -ffp-contract=off -fsanitize=address,undefined -fno-gpu-sanitize \
      -o /tmp/librk_asan.so rappas_amd/csrc/rk_engine.hip rappas_amd/csrc/rk_pack_host.cpp
LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=halt_on_error=1 RK_LIB=/tmp/librk_asan.so \
      python -m pytest tests/test_cabi.py tests/test_db_image.py -q -x -k "validate or argument or image_written or damaged" -m "not gpu" -p no:cacheprovider
gcc -O1 -g -std=c99 -fPIC -ffp-contract=off -fsanitize=address,undefined -shared -o /tmp/liboracle_asan.so oracle/rappas_oracle.c oracle/rappas_build_oracle.c -lm
LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=halt_on_error=1 RO_LIB=/tmp/liboracle_asan.so \
      python -m pytest tests/test_oracle_golden.py tests/test_oracle_build.py -q -x -p no:cacheprovider
echo "asan/ubsan engine host code + oracles: OK"
# the native host side (rk_hostio.hpp through rk_place's device-free modes) under the same sanitizers
g++ -O1 -g -std=c++17 -fsanitize=address,undefined -fno-sanitize-recover=undefined -o /tmp/rk_place_asan rappas_amd/csrc/host/rk_place_main.cpp \
    -Lrappas_amd -lrappas_place -Wl,-rpath,$PWD/rappas_amd
T=$(mktemp -d)
printf '((A:0.1,B:0.2)C:0.3,(D:0.4,E:1e-3)F:1234.5)R;' > $T/t.nwk
printf '>r1 d\nACGT\nAC-GT\n>r2\nACGTACGT\n>r3 x\nACGTACG-T\n' > $T/q.fa
python - "$T" <<'PY'
import sys
from rappas_amd import hostio, synth
open(sys.argv[1] + "/db.json", "w").write(hostio.dump_jsondb(synth.make_db(4, 5, 9, 150, 700, seed=5), synth.make_newick(9, seed=1)))
from tests import test_uniondb as TU
blob = TU.toy_stream()
open(sys.argv[1] + "/toy.union", "wb").write(blob)
for cut in range(0, len(blob), 7):  # every seventh truncation of the stream: the reader has to stop with an error, not read past it
    open(sys.argv[1] + f"/cut{cut}.union", "wb").write(blob[:cut])
for name, h in TU.hostile_streams().items():  # a descriptor that is its own super class, arrays nested 5000 deep
    open(sys.argv[1] + f"/flip_{name}.union", "wb").write(h)
import random
rnd = random.Random(1)
for i in range(60):  # single-byte corruptions
    b = bytearray(blob)
    b[rnd.randrange(len(b))] = rnd.randrange(256)
    open(sys.argv[1] + f"/flip{i}.union", "wb").write(bytes(b))
PY
python - "$T" <<'PY'
import sys
import numpy as np
sys.path.insert(0, ".")
from tests import test_host_cpp as TH
from rappas_amd import synth
open(sys.argv[1] + "/messy.fa", "w").write(TH._messy_fasta(np.random.default_rng(2), 400, weird_names=True))
open(sys.argv[1] + "/t120.nwk", "w").write(synth.make_newick(120, seed=3))
PY
# (round 4) the all-threads host path: scan + dedup, both jplace writers on made-up placements, the image written by the tool
for args in "--emit-tree $T/t.nwk" "--dedup $T/q.fa" "--load-jsondb $T/db.json" "--load-uniondb $T/toy.union" "--format-float 0.1" "--format-double 1e-300" "--md5 abc" \
            "--threads 3 --dedup-fast $T/messy.fa" "--threads 1 --md5-dedup --dedup-fast $T/messy.fa" "--uniondb-stats $T/toy.union" \
            "--threads 5 --keep-at-most 7 --write-selftest $T/messy.fa $T/t120.nwk $T/a.jplace $T/b.jplace 1" \
            "--threads 2 --keep-at-most 3 --guppy-compat --write-selftest $T/q.fa $T/t120.nwk $T/a.jplace $T/b.jplace 2" \
            "--jsondb $T/db.json --save-dbimage $T/db.rkimg"; do
  ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=halt_on_error=1 /tmp/rk_place_asan $args > /dev/null
done
for f in $T/cut*.union $T/flip*.union; do  # exit 0 (still a valid stream) or 1 (rejected); anything else is a sanitizer report or a crash
  rc=0; ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=halt_on_error=1 /tmp/rk_place_asan --load-uniondb $f > /dev/null 2>$T/err || rc=$?
  if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then echo "union reader: $f -> exit $rc"; cat $T/err | head -20; exit 1; fi
done
rm -rf $T
echo "asan/ubsan native host side: OK"
