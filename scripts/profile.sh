#!/bin/bash
# rocprofv3 passes for the placement kernel (run on the GPU box through gpurun).
# usage: scripts/profile.sh <tag> [bench args...]   -> gpurun_out/prof_<tag>/
set -u
TAG=${1:-r02}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
STEPS=${PROF_TRACE_STEPS:-20}
BENCH="python3 $ROOT/bench.py --steps 3 --warmup 1 --verify 0 --no-cpu-baseline --no-pcie $*"
run() { # name, rocprof args...
  local name=$1; shift
  timeout -k 10 280 rocprofv3 "$@" --output-format csv -d "$OUT/$name" -- $BENCH > "$OUT/$name.log" 2>&1 || { echo "pass $name failed"; tail -5 "$OUT/$name.log"; return 1; }
}
# the timing pass runs more steps than the counter passes: its average has to agree with the bench line's kernel_ms
BENCH="python3 $ROOT/bench.py --steps $STEPS --warmup 2 --verify 0 --no-cpu-baseline --no-pcie $*" run trace --kernel-trace --stats &&
{ grep -h "^{" "$OUT/trace.log" | tail -1 > "$OUT/../prof_${TAG}_bench_line.json"; true; } &&
run pmc1 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU &&
run pmc2 --kernel-trace --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_WAVES &&
run pmc3 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE &&
run pmc4 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
if [ -n "${PROF_PREFIX:-}" ]; then  # summarise, keep only the summaries (raw traces exceed the 64 MiB merge limit)
  python3 "$ROOT/scripts/prof_summary.py" "$OUT" "$ROOT/gpurun_out/$PROF_PREFIX"
  rm -rf "$OUT"
else
  find "$OUT" -name "*.csv" | head -30
fi
