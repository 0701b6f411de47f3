#!/bin/bash
# Timing of one-off compile-time variants of the placement kernel: FLAGS="-DX=1" scripts/tune_flag.sh [bench args]
cd "$(dirname "$0")/.."
tag=$(echo "$FLAGS" | tr -c 'A-Za-z0-9' '_')
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off $FLAGS -o /tmp/librk_flag_$tag.so rappas_amd/csrc/rk_engine.hip rappas_amd/csrc/rk_pack_host.cpp || exit 1
RK_LIB=/tmp/librk_flag_$tag.so timeout -k 5 120 python bench.py --steps 5 --warmup 1 --verify 2000 --no-cpu-baseline "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('flags=$FLAGS', round(d['roofline']['kernel_ms'],2), 'ms', round(d['value']/1e6,1), 'Mreads/s')"
