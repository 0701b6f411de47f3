"""Developer measurement: placement rate against the tree size (C2-like DB: DNA k=10, 786432 keys, 1e7 entries, 150 bp)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rappas_amd as ra
from rappas_amd import synth

sizes = [int(x) for x in sys.argv[1:]] or [999, 1999, 3999, 7999, 8193, 12001]
n = 2_000_000
for nb in sizes:
    sdb = synth.make_db(4, 10, nb, 786432, 10_000_000, seed=42)
    db = ra.PhyloKmerDB.from_synth(sdb)
    pp = ra.PlacementProcess(db)
    wpr = db.packed_words(150)
    packed = torch.randint(-2**31, 2**31, (n, wpr), dtype=torch.int64, device="cuda").to(torch.int32)
    packed[:, wpr - 1] &= (1 << (300 - 32 * (wpr - 1))) - 1
    for lanes in ([0] if nb > 8192 else [0, 16, 32, 64]):
        try:
            db.set_lanes_per_read(lanes)
            out = pp.place_packed(packed, fixed_len=150); torch.cuda.synchronize()
        except Exception as e:
            print(f"n_branches={nb:6d} lanes={lanes}: {str(e)[:80]}", flush=True)
            continue
        t = time.time()
        for _ in range(3):
            pp.place_packed(packed, fixed_len=150, out=out)
        torch.cuda.synchronize()
        dt = (time.time() - t) / 3
        print(f"n_branches={nb:6d} lanes={lanes:2d}: {n / dt / 1e6:8.1f} Mreads/s   [{db.kernel_name()}]", flush=True)
    db.close()
