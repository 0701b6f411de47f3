cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
R=$PWD
cd /tmp
for w in sparse dense; do
  rm -rf $R/gpurun_out/dv
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/dv -- python3 $R/scripts/dev/density_verdict_check.py $w > $R/gpurun_out/dv_$w.log 2>&1 || { tail -3 $R/gpurun_out/dv_$w.log; exit 1; }
  echo "== $w batch"
  python3 - $R/gpurun_out/dv <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "place_" in row["Name"] or "compact" in row["Name"]:
            print(f'   {row["Name"][:70]:70s} calls {row["Calls"]:>3s} avg {float(row["AverageNs"]) / 1e3:9.1f} us')
PY
done
rm -rf $R/gpurun_out/dv
