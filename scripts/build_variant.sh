#!/bin/bash
# Developer helper: rappas_amd/variants/librk_<name>.so = the engine compiled with extra -D flags (A/B runs: RK_LIB=<path>).
# usage: scripts/build_variant.sh <name> [-DRK_...=...]...
cd "$(dirname "$0")/.."
name="$1"; shift
mkdir -p rappas_amd/variants build/variants
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DRK_DEV_KNOBS -Wall -Wno-unused-function "$@" \
  -Rpass-analysis=kernel-resource-usage -c -o build/variants/rk_engine_$name.o rappas_amd/csrc/rk_engine.hip 2> build/variants/$name.resources.txt || { tail -20 build/variants/$name.resources.txt; exit 1; }
[ -f build/obj/rk_build_dev.o ] || python -m rappas_amd.build
hipcc --offload-arch=gfx950 -shared -fPIC -o rappas_amd/variants/librk_$name.so build/variants/rk_engine_$name.o build/obj/rk_build_dev.o build/obj/rk_pack_host.o
echo "built rappas_amd/variants/librk_$name.so (resource usage: build/variants/$name.resources.txt)"
