import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import rappas_amd as ra
from rappas_amd import synth
n = 500_000
for nb in (9001, 12001, 15999):
    for mean in (30, 45, 60):
        sdb = synth.make_db(4, 9, nb, 262144, 262144 * mean, seed=42)
        db = ra.PhyloKmerDB.from_synth(sdb)
        pp = ra.PlacementProcess(db)
        wpr = db.packed_words(150)
        packed = torch.randint(-2**31, 2**31, (n, wpr), dtype=torch.int64, device="cuda").to(torch.int32)
        packed[:, wpr - 1] &= (1 << (300 - 32 * (wpr - 1))) - 1
        out = pp.place_packed(packed, fixed_len=150); torch.cuda.synchronize()
        t = time.time()
        for _ in range(3):
            pp.place_packed(packed, fixed_len=150, out=out)
        torch.cuda.synchronize()
        print(f"n_branches={nb:6d} all k-mers present, mean row {mean:3d}: {n / ((time.time() - t) / 3) / 1e6:7.1f} Mreads/s   [{db.kernel_name()[:44]}]", flush=True)
        db.close()
