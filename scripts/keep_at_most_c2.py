import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import rappas_amd as ra
from rappas_amd import synth
n = 4_000_000
sdb = synth.make_config_db("C2")
db = ra.PhyloKmerDB.from_synth(sdb)
pp = ra.PlacementProcess(db)
wpr = db.packed_words(150)
packed = torch.randint(-2**31, 2**31, (n, wpr), dtype=torch.int64, device="cuda").to(torch.int32)
packed[:, wpr - 1] &= (1 << (300 - 32 * (wpr - 1))) - 1
for K in (7, 8, 9, 10, 12, 14, 16):
    out = pp.place_packed(packed, fixed_len=150, keepAtMost=K); torch.cuda.synchronize()
    t = time.time()
    for _ in range(3):
        pp.place_packed(packed, fixed_len=150, keepAtMost=K, out=out)
    torch.cuda.synchronize()
    print(f"C2 keep_at_most={K:2d} {os.environ.get('RK_LIB','product')[-12:]}: {n / ((time.time() - t) / 3) / 1e6:7.1f} Mreads/s", flush=True)
