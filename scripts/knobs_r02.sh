#!/bin/bash
# Developer sweep (round 2): occupancy / list-capacity / ring-depth sensitivity of the C2 kernel. Variant libraries are prebuilt
# into rappas_amd/variants/ on the build host (they travel to the GPU box with the snapshot).
cd "$(dirname "$0")/.."
run() {  # label, env assignments...
  local label="$1"; shift
  env "$@" timeout -k 5 120 python bench.py --steps 10 --warmup 2 --verify 500 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label', round(d['roofline']['kernel_ms'],3), 'ms', round(d['value']/1e6,1), 'Mreads/s frac', round(d['roofline']['frac'],4), d['config']['kernel'])"
}
run base X=1
for w in 7 6 4; do run "waves/CU=$w" RK_WAVES_PER_CU=$w; done
for c in 100 120 200; do run "list_cap=$c" RK_LIST_CAP=$c; done
for u in 12 16; do [ -f rappas_amd/variants/librk_ring_$u.so ] && run "ring=$u" RK_LIB=$PWD/rappas_amd/variants/librk_ring_$u.so; done
run base-again X=1
