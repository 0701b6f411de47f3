# developer helper (GPU box): stamped builds of place_hash64_kernel with the ring / entries-per-lane variants named on the command line
# usage: bash scripts/run_stamps_hash.sh OUT "U NPL [extra -D flags]" ...
set -e
cd ${GRAFT_REPO_ROOT:-.}
out=$1; shift
mkdir -p rappas_amd/variants gpurun_out
: > $out
for v in "$@"; do
  set -- $v
  u=$1; npl=$2; shift 2
  so=rappas_amd/variants/librk_stamps_${u}_${npl}$(echo "$*" | tr -d ' =-').so
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -DRK_DEV_KNOBS -DRK_STAMPS -DRK_HRING=$u -DRK_HNPL=$npl "$@" -o $so rappas_amd/csrc/rk_engine.hip rappas_amd/csrc/rk_pack_host.cpp
  echo "=== RK_HRING=$u RK_HNPL=$npl $*" >> $out
  RK_STAMPS_LIB=$so python scripts/stamps_hash.py --branches=7999 --branches=65535 2>/dev/null >> $out
  RK_STAMPS_LIB=$so python scripts/stamps_hash.py --clade --branches=19999 2>/dev/null >> $out
done
cat $out
