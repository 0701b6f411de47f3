"""Summarise rocprofv3 CSV output (kernel stats + PMC counters) for the placement kernels.
usage: prof_summary.py <prof_dir> [<out_prefix>]  -> <out_prefix>_kernel_stats.csv, _pmc_summary.txt, _pmc_traffic.json"""
import collections
import csv
import glob
import json
import os
import sys

out = sys.argv[1]
prefix = sys.argv[2] if len(sys.argv) > 2 else None


def rows(pat):
    for f in glob.glob(os.path.join(out, pat), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                yield f, r


lines = []
stats = [r for _, r in rows("trace/**/*kernel_stats.csv")]
for r in stats:
    if "rk::" in r.get("Name", "") or float(r.get("Percentage", 0) or 0) > 2:
        lines.append(f"STAT {r.get('Name', '')[:100]} calls={r.get('Calls')} avg_ns={r.get('AverageNs')} min_ns={r.get('MinNs')} max_ns={r.get('MaxNs')} pct={r.get('Percentage')}")
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f, r in rows("pmc*/**/*counter_collection.csv"):
    name = r.get("Kernel_Name", "")
    if "rk::" not in name:
        continue
    agg[name[:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    lines.append("KERNEL " + k)
    for c, v in sorted(d.items()):
        lines.append(f"   {c:28s} launches={len(v)} mean_per_launch={sum(v)/len(v):.6g}")
print("\n".join(lines))
if prefix:
    if stats:
        with open(prefix + "_kernel_stats.csv", "w", newline="") as fh:
            w = csv.DictWriter(fh, fieldnames=list(stats[0].keys()))
            w.writeheader()
            w.writerows(stats)
    with open(prefix + "_pmc_summary.txt", "w") as fh:
        fh.write("\n".join(lines) + "\n")
    # the dominant placement kernel (the one that fetched most: place_packed16s_kernel hands a few tiles to a second launch)
    cands = [k for k, d in agg.items() if "rk::place_" in k and "place_ascii" not in k and "FETCH_SIZE" in d and "WRITE_SIZE" in d]
    cands.sort(key=lambda k: -sum(agg[k]["FETCH_SIZE"]) / len(agg[k]["FETCH_SIZE"]))
    for k, d in [(k, agg[k]) for k in cands[:1]]:
        if True:
            fetch = sum(d["FETCH_SIZE"]) / len(d["FETCH_SIZE"])
            write = sum(d["WRITE_SIZE"]) / len(d["WRITE_SIZE"])
            meta = json.loads(os.environ.get("PROF_META", "{}"))
            if not meta:  # what was profiled: from the bench line profile.sh keeps of its timing pass
                try:
                    bl = json.loads(open(os.path.join(os.path.dirname(out.rstrip("/")), os.path.basename(out.rstrip("/")) + "_bench_line.json")).read())
                    meta = {"config": bl["config"]["workload"].split(":")[0], "reads": bl["config"]["reads_per_gpu"]}
                except Exception:
                    meta = {}
            miss = sum(d["TCC_MISS_sum"]) / len(d["TCC_MISS_sum"]) if "TCC_MISS_sum" in d else None
            meta.update({"kernel": k, "FETCH_SIZE_KB_per_launch": fetch, "WRITE_SIZE_KB_per_launch": write,
                         "TCC_MISS_per_launch": miss,
                         "fetch_bytes_per_L2_miss_as_reported": (fetch * 1024 / miss) if miss else None,
                         "hbm_bytes_per_launch": (2 * fetch + write) * 1024,
                         "note": "FETCH_SIZE/WRITE_SIZE are in KiB per launch (separate --pmc passes). Rows are read as aligned "
                                 "128-byte units (16 lanes x 8 B), and FETCH_SIZE comes out at ~64 B per L2 miss, i.e. 128-byte "
                                 "requests tallied at 64 B: the x2 correction of MI355X_MICROARCH.md (HBM section) is applied to "
                                 "FETCH_SIZE, WRITE_SIZE is taken as is. These are fabric-side counters: the C2 row blob (138 MB) "
                                 "sits in the 256 MB Infinity Cache and its hits are included, so this is L2-miss traffic, an upper "
                                 "bound of what reaches HBM. It exceeds the algorithmic bytes because rows are padded to whole "
                                 "128-byte units (the request COUNT, not the byte count, is what bounds this gather: DESIGN.md section 5)."})
            with open(prefix + "_pmc_traffic.json", "w") as fh:
                json.dump(meta, fh, indent=1)
