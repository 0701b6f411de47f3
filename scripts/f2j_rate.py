"""Developer measurement: the fasta_to_jplace leg of bench.py on its own (C2's database, --reads N uniform reads), by thread count."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import rappas_amd as ra
from rappas_amd import synth
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
sdb = synth.make_config_db("C2")
db = ra.PhyloKmerDB.from_synth(sdb)
seq, off = synth.make_reads(4, n, 150, seed=1)
print(f"host: {os.cpu_count()} cpus, {len(os.sched_getaffinity(0))} usable", flush=True)
for th in [int(x) for x in sys.argv[2:]] or [32]:
    os.sched_setaffinity(0, os.sched_getaffinity(0))
    orig = os.sched_getaffinity
    os.sched_getaffinity = lambda pid, _th=th: set(range(_th))  # (the leg sizes its team from the usable CPUs)
    r = bench.fasta_to_jplace_leg(ra, db, seq, off, n, 150, sdb.n_branches, 7)
    os.sched_getaffinity = orig
    p = r["passes"]
    print(f"{th:3d} threads: {r['value'] / 1e6:6.2f} Mreads/s   scan {p['scan_s'] * 1e3:6.1f}  dedup {p['dedup_s'] * 1e3:6.1f}  gather {p['gather_s'] * 1e3:6.1f}  place {p['place_s'] * 1e3:6.1f} (waiting for the engine's warm-up {p['place_wait_for_warm_up_s'] * 1e3:5.1f} of its {p['engine_warm_up_s'] * 1e3:5.1f})  "
          f"write {p['write_s'] * 1e3:6.1f} (format {p['write_format_s'] * 1e3:6.1f}, io {p['write_io_s'] * 1e3:6.1f}) ms   {p['fasta_bytes'] / 1e6:.0f} MB in, {p['jplace_bytes'] / 1e6:.0f} MB out", flush=True)
