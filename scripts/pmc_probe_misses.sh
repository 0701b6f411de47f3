#!/bin/bash
# TCC misses of the C2 launch with every row load redirected to one cached line (RK_ABLATE=16): what is left are probes, records, results
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
cd /tmp
export RK_LIB=$ROOT/rappas_amd/variants/librk_abl16.so
timeout -k 10 280 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d /tmp/pmc_abl -- python3 $ROOT/bench.py --steps 3 --warmup 1 --verify 0 --no-cpu-baseline --no-pcie > /tmp/pmc_abl.log 2>&1 || { tail -5 /tmp/pmc_abl.log; exit 1; }
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("/tmp/pmc_abl/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "place_packed16_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    print(k, len(v), sum(v) / len(v), "per read:", sum(v) / len(v) / 1e7)
PY
