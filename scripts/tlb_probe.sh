# developer helper (GPU box): address-translation counters of place_wg_kernel on C5-shaped workloads of growing footprint
# (C5s = 1 GB of rows, C5m = 12 GB, C5 = 200 GB) -> gpurun_out/r04_tlb_probe.txt
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
R=$PWD
cd /tmp
: > $R/gpurun_out/r04_tlb_probe.txt
for c in "$@"; do
  rm -rf $R/gpurun_out/tlb
  timeout -k 10 500 rocprofv3 --kernel-trace --pmc TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/tlb -- python3 $R/bench.py --config $c --steps 2 --warmup 1 --verify 0 --no-cpu-baseline --no-pcie > $R/gpurun_out/tlb_$c.log 2>&1 || { tail -5 $R/gpurun_out/tlb_$c.log; exit 1; }
  python3 - $R/gpurun_out/tlb $c $R/gpurun_out/tlb_$c.log >> $R/gpurun_out/r04_tlb_probe.txt <<'PY'
import csv, glob, sys, collections, json
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"][:40]
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"]); n[(k, row["Counter_Name"])] += 1
line = [l for l in open(sys.argv[3]) if l.startswith("{")]
d = json.loads(line[-1]) if line else {}
for k, dd in acc.items():
    if "place_wg" in k:
        v = {c: x / n[(k, c)] for c, x in dd.items()}
        req, hit, miss = v.get("TCP_UTCL1_REQUEST_sum", 0), v.get("TCP_UTCL1_TRANSLATION_HIT_sum", 0), v.get("TCP_UTCL1_TRANSLATION_MISS_sum", 0)
        print(f"{sys.argv[2]:4s} reads/s {d.get('value', 0):.3g} frac {d.get('roofline', {}).get('frac', 0):.3f} | per launch: UTCL1 requests {req:.3g}, hits {hit:.3g}, misses {miss:.3g} ({100 * miss / max(1, req):.1f} % of requests), "
              f"stalled on UTCL2 credits {v.get('TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum', 0):.3g} cycles, GRBM_GUI_ACTIVE {v.get('GRBM_GUI_ACTIVE', 0):.3g} cycles")
PY
done
rm -rf $R/gpurun_out/tlb
cat $R/gpurun_out/r04_tlb_probe.txt
