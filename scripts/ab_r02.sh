#!/bin/bash
# Developer A/B on one box: product library vs every prebuilt variant in rappas_amd/variants/ (interleaved, two rounds).
cd "$(dirname "$0")/.."
run() {
  local label="$1"; shift
  env "$@" timeout -k 5 120 python bench.py --steps 15 --warmup 3 --verify 1000 --no-cpu-baseline ${BENCH_ARGS} 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label', round(d['roofline']['kernel_ms'],3), 'ms', round(d['value']/1e6,1), 'Mreads/s frac', round(d['roofline']['frac'],4))"
}
for round in 1 2; do
  run product X=1
  for v in rappas_amd/variants/*.so; do [ -f "$v" ] && run "$(basename $v)" RK_LIB=$PWD/$v; done
done
