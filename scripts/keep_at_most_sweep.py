"""Developer measurement: placement rate against keep_at_most on mid-size trees (C2-like DB).  RK_NO_WINDOW=1 forces the dense kernels."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rappas_amd as ra
from rappas_amd import synth
n = 2_000_000
mode = "dense   " if os.environ.get("RK_NO_WINDOW") else "windowed"
for nb in [int(x) for x in sys.argv[1:]] or (3999, 7999):
    sdb = synth.make_db(4, 10, nb, 786432, 10_000_000, seed=42)
    db = ra.PhyloKmerDB.from_synth(sdb)
    pp = ra.PlacementProcess(db)
    wpr = db.packed_words(150)
    packed = torch.randint(-2**31, 2**31, (n, wpr), dtype=torch.int64, device="cuda").to(torch.int32)
    packed[:, wpr - 1] &= (1 << (300 - 32 * (wpr - 1))) - 1
    for K in (7, 8, 9, 12, 14, 16):
        out = pp.place_packed(packed, fixed_len=150, keepAtMost=K); torch.cuda.synchronize()
        t = time.time()
        for _ in range(3):
            pp.place_packed(packed, fixed_len=150, keepAtMost=K, out=out)
        torch.cuda.synchronize()
        print(f"n_branches={nb} keep_at_most={K:2d} {mode}: {n / ((time.time() - t) / 3) / 1e6:7.1f} Mreads/s", flush=True)
    db.close()
