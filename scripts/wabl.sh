#!/bin/bash
# timing-only ablations of the windowed kernel (outputs wrong by construction): 512 no accumulate, 1024 select only in the last window,
# 2048 no compaction + accumulate, 3072 = 1024 + 2048
cd "$(dirname "$0")/.."
for nb in 1999 3999; do
  python scripts/tree_size_sweep.py $nb 2>/dev/null | grep "lanes= 0"
  for m in 512 2048; do
    RK_LIB=$PWD/rappas_amd/variants/librk_wabl_$m.so python scripts/tree_size_sweep.py $nb 2>/dev/null | grep "lanes= 0" | sed "s/^/ablate=$m /"
  done
done
