#!/bin/bash
# timing-only variants of the windowed kernel (ablations give wrong outputs by construction)
cd "$(dirname "$0")/.."
for nb in 3999 7999; do
  python scripts/tree_size_sweep.py $nb 2>/dev/null | grep "lanes= 0"
  for v in rappas_amd/variants/librk_*.so; do
    RK_LIB=$PWD/$v python scripts/tree_size_sweep.py $nb 2>/dev/null | grep "lanes= 0" | sed "s/^/$(basename $v) /" | cut -c1-90
  done
done
