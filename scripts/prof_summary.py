"""Summarise rocprofv3 CSV output (kernel stats + PMC counters) for the placement kernel."""
import csv, glob, os, sys, collections
out = sys.argv[1]
def rows(pat):
    for f in glob.glob(os.path.join(out, pat), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                yield f, r
# kernel stats
for f, r in rows("trace/**/*kernel_stats.csv"):
    if "place" in r.get("Name", "") or float(r.get("Percentage", 0) or 0) > 2:
        print("STAT", r.get("Name", "")[:90], "calls", r.get("Calls"), "avg_ns", r.get("AverageNs"), "pct", r.get("Percentage"))
# counters
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f, r in rows("pmc*/**/*counter_collection.csv"):
    name = r.get("Kernel_Name", "")
    if "place" not in name: continue
    agg[name[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print("KERNEL", k)
    for c, v in sorted(d.items()):
        print(f"   {c:28s} n={len(v)} mean={sum(v)/len(v):.4g}")
