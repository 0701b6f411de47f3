# developer helper (GPU box): instruction-cache counters of place_wg_kernel on 15 999 branches, rows of 400
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp RK_SIZES=15999
R=$PWD
cd /tmp
rocprofv3 -L 2>/dev/null | grep -i -o "SQC_ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQC_INST[A-Z_]*" | sort -u > $R/gpurun_out/icache_counters.txt
cat $R/gpurun_out/icache_counters.txt
timeout -k 10 280 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/icache -- python3 $R/scripts/long_rows_big_tree.py 400 > $R/gpurun_out/icache.log 2>&1 || { tail -5 $R/gpurun_out/icache.log; exit 1; }
python3 - $R/gpurun_out/icache <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"][:60]
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"]); n[(k, row["Counter_Name"])] += 1
for k, d in acc.items():
    if "place_" in k:
        print(k, {c: round(v / n[(k, c)]) for c, v in d.items()})
PY
rm -rf $R/gpurun_out/icache
