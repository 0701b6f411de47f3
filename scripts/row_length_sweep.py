"""Developer measurement: placement rate against the mean row length on a mid-size tree (RK_NO_WINDOW=1 forces the dense kernels)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rappas_amd as ra
from rappas_amd import synth
mode = "dense   " if os.environ.get("RK_NO_WINDOW") else "auto    "
n = 1_000_000
for nb in [int(x) for x in os.environ.get("RK_SWEEP_BRANCHES", "3999,7999").split(",")]:
    for mean in [int(x) for x in sys.argv[1:]] or [13, 40, 100, 250]:
        keys = 200_000
        sdb = synth.make_db(4, 10, nb, keys, keys * mean, seed=42)
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
        dt = (time.time() - t) / 3
        print(f"n_branches={nb:6d} mean row {mean:4d} {mode}: {n / dt / 1e6:7.1f} Mreads/s   [{db.kernel_name()[:48]}]", flush=True)
        db.close()
