#!/bin/bash
cd "$(dirname "$0")/.."
for u in ${RING_SET:-8 12 16 24}; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -DRK_DEV_KNOBS -DRK_RING=$u -o /tmp/librk_ring_$u.so rappas_amd/csrc/rk_engine.hip rappas_amd/csrc/rk_pack_host.cpp || exit 1
  RK_LIB=/tmp/librk_ring_$u.so timeout -k 5 120 python bench.py --steps 5 --warmup 1 --verify 2000 --no-cpu-baseline "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ring=$u', round(d['roofline']['kernel_ms'],2), 'ms', round(d['value']/1e6,1), 'Mreads/s', d['config']['kernel'])"
done
