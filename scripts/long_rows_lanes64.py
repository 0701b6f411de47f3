"""Developer measurement: what the engine picks (lanes=0) against forced lane-group widths on long-row shapes (see long_rows_big_tree.py)."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import rappas_amd as ra
from rappas_amd import synth
n = 300_000
for nb in [int(x) for x in os.environ.get("RK_SIZES", "9001,13001").split(",")]:
    for mean in [int(x) for x in sys.argv[1:]] or [70, 150, 300]:
        keys = 65536
        sdb = synth.make_db(4, 9, nb, keys, keys * mean, seed=42)
        db = ra.PhyloKmerDB.from_synth(sdb)
        pp = ra.PlacementProcess(db)
        wpr = db.packed_words(150)
        packed = torch.randint(-2**31, 2**31, (n, wpr), dtype=torch.int64, device="cuda").to(torch.int32)
        packed[:, wpr - 1] &= (1 << (300 - 32 * (wpr - 1))) - 1
        for lanes in [int(x) for x in os.environ.get("RK_LANES", "0,64").split(",")]:
            db.set_lanes_per_read(lanes)
            out = pp.place_packed(packed, fixed_len=150); torch.cuda.synchronize()
            t = time.time()
            for _ in range(3):
                pp.place_packed(packed, fixed_len=150, out=out)
            torch.cuda.synchronize()
            print(f"n_branches={nb:6d} mean row {mean:5d} lanes={lanes:2d}: {n / ((time.time() - t) / 3) / 1e6:7.1f} Mreads/s   [{db.kernel_name()[:44]}]", flush=True)
        db.close()
