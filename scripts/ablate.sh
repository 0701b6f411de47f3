#!/bin/bash
# Timing-only ablation builds of the placement kernel (outputs are wrong by construction; never shipped).
cd "$(dirname "$0")/.."
for m in ${ABLATE_SET:-0 1 2 4 3 7}; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -DRK_DEV_KNOBS -DRK_ABLATE=$m -o /tmp/librk_ablate_$m.so rappas_amd/csrc/rk_engine.hip rappas_amd/csrc/rk_pack_host.cpp || exit 1
  RK_LIB=/tmp/librk_ablate_$m.so timeout -k 5 120 python bench.py --steps 5 --warmup 1 --verify 0 --no-cpu-baseline "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ablate=$m', round(d['roofline']['kernel_ms'],2), 'ms', round(d['value']/1e6,1), 'Mreads/s')"
done
