set -e
cd $GRAFT_REPO_ROOT
for c in ${PROF_CONFIGS:-C2 T64k L16k}; do
  lc=$(echo $c | tr A-Z a-z)
  PROF_PREFIX=r04_${lc} bash scripts/profile.sh $lc --config $c --no-clade > gpurun_out/r04_prof_${lc}.log 2>&1 || { tail -5 gpurun_out/r04_prof_${lc}.log; exit 1; }
  cp gpurun_out/prof_${lc}_bench_line.json gpurun_out/r04_${lc}_profiled_bench_line.json 2>/dev/null || true
  echo "profiled $c"
done
ls gpurun_out | grep r04_ | head -40
