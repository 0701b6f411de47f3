#!/bin/bash
# Host-side AddressSanitizer + UBSan pass (CPU only; GPU sanitizers are not available on this pool):
#   * the engine's host code (argument validation + HBM image construction, rk_db_validate) built with
#     -fsanitize=address,undefined -fno-gpu-sanitize, exercised by tests/test_cabi.py -k validate
#   * the C oracles built with the same sanitizers, exercised by tests/test_oracle_golden.py and tests/test_oracle_build.py
set -e
cd "$(dirname "$0")/.."
RT=$(/opt/rocm/lib/llvm/bin/clang --print-file-name=libclang_rt.asan-x86_64.so)
hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -shared -ffp-contract=off -fsanitize=address,undefined -fno-gpu-sanitize \
      -o /tmp/librk_asan.so rappas_amd/csrc/rk_engine.hip
LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=halt_on_error=1 RK_LIB=/tmp/librk_asan.so \
      python -m pytest tests/test_cabi.py -q -x -k "validate or argument" -p no:cacheprovider
gcc -O1 -g -std=c99 -fPIC -ffp-contract=off -fsanitize=address,undefined -shared -o /tmp/liboracle_asan.so oracle/rappas_oracle.c oracle/rappas_build_oracle.c -lm
LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=halt_on_error=1 RO_LIB=/tmp/liboracle_asan.so \
      python -m pytest tests/test_oracle_golden.py tests/test_oracle_build.py -q -x -p no:cacheprovider
echo "asan/ubsan host pass: OK"
