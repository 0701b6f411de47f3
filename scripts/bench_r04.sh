# developer helper (GPU box): the round's bench lines, one file per configuration -> gpurun_out/r04_bench_<config>.json
set -e
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
for c in "$@"; do
  lc=$(echo $c | tr A-Z a-z)
  if [ $c = C2 ]; then python bench.py > gpurun_out/r04_bench_${lc}.json 2> gpurun_out/r04_bench_${lc}.err
  else python bench.py --config $c > gpurun_out/r04_bench_${lc}.json 2> gpurun_out/r04_bench_${lc}.err; fi
  python - gpurun_out/r04_bench_${lc}.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print(f'{sys.argv[1]}: {d["value"]:.4g} {d["unit"]} frac {r["frac"]:.3f} kernel_ms {r.get("kernel_ms")} clade {d.get("clade", {}).get("value") if isinstance(d.get("clade"), dict) else d.get("clade")}', flush=True)
PY
done
