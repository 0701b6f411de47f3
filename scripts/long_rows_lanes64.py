import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import rappas_amd as ra
from rappas_amd import synth
n = 300_000
for nb in (9001, 13001):
    for mean in (70, 150, 300):
        keys = 65536
        sdb = synth.make_db(4, 9, nb, keys, keys * mean, seed=42)
        db = ra.PhyloKmerDB.from_synth(sdb)
        pp = ra.PlacementProcess(db)
        wpr = db.packed_words(150)
        packed = torch.randint(-2**31, 2**31, (n, wpr), dtype=torch.int64, device="cuda").to(torch.int32)
        packed[:, wpr - 1] &= (1 << (300 - 32 * (wpr - 1))) - 1
        for lanes in (0, 64):
            db.set_lanes_per_read(lanes)
            out = pp.place_packed(packed, fixed_len=150); torch.cuda.synchronize()
            t = time.time()
            for _ in range(3):
                pp.place_packed(packed, fixed_len=150, out=out)
            torch.cuda.synchronize()
            print(f"n_branches={nb:6d} mean row {mean:5d} lanes={lanes:2d}: {n / ((time.time() - t) / 3) / 1e6:7.1f} Mreads/s   [{db.kernel_name()[:44]}]", flush=True)
        db.close()
