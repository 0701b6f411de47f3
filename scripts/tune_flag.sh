#!/bin/bash
# usage: scripts/tune_flag.sh "<-Dflags A>" "<-Dflags B>" ... ; builds one variant per argument and benches it
cd "$(dirname "$0")/.."
i=0
for f in "$@"; do
  i=$((i+1))
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off $f -o /tmp/librk_flag_$i.so rappas_amd/csrc/rk_engine.hip || exit 1
  RK_LIB=/tmp/librk_flag_$i.so timeout -k 5 120 python bench.py --steps 5 --warmup 1 --verify 2000 --no-cpu-baseline $BENCH_ARGS 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$f]', round(d['roofline']['kernel_ms'],2), 'ms', round(d['value']/1e6,1), 'Mreads/s')"
done
