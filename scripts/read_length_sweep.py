"""Developer measurement: placement rate against the read length (C2-like DB; 999 and 3 999 branches), k-mers/s for comparison."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rappas_amd as ra
from rappas_amd import synth

lengths = [int(x) for x in sys.argv[1:]] or [100, 150, 250, 256, 300, 450, 600]
for nb in [int(x) for x in os.environ.get("RK_SWEEP_BRANCHES", "999,3999").split(",")]:
    sdb = synth.make_db(4, 10, nb, 786432, 10_000_000, seed=42)
    db = ra.PhyloKmerDB.from_synth(sdb)
    pp = ra.PlacementProcess(db)
    for L in lengths:
        n = int(3e8 / L)
        wpr = db.packed_words(L)
        packed = torch.randint(-2**31, 2**31, (n, wpr), dtype=torch.int64, device="cuda").to(torch.int32)
        if (2 * L) % 32:
            packed[:, wpr - 1] &= (1 << (2 * L - 32 * (wpr - 1))) - 1
        out = pp.place_packed(packed, fixed_len=L); torch.cuda.synchronize()
        t = time.time()
        for _ in range(3):
            pp.place_packed(packed, fixed_len=L, out=out)
        torch.cuda.synchronize()
        dt = (time.time() - t) / 3
        print(f"n_branches={nb:5d} read length {L:4d} ({wpr:2d} words): {n / dt / 1e6:7.1f} Mreads/s  {n * (L - 9) / dt / 1e9:6.1f} G k-mers/s", flush=True)
    db.close()
