"""Developer measurement: placement rate against the database size at a fixed small tree (999 branches, rows of ~13 entries)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rappas_amd as ra
from rappas_amd import synth

n = 2_000_000
for k, n_keys in [(10, 786432), (11, 3145728), (12, 12582912), (13, 50331648)]:
    sdb = synth.make_db(4, k, 999, n_keys, int(n_keys * 12.7), seed=42)
    t0 = time.time()
    db = ra.PhyloKmerDB.from_synth(sdb)
    t_create = time.time() - t0
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
    print(f"k={k} keys={n_keys} entries={sdb.n_entries} rows={db.info.rows_bytes/1e6:.0f} MB table={db.info.table_bytes/1e6:.1f} MB "
          f"(rk_db_create {t_create:.1f} s): {n / dt / 1e6:8.1f} Mreads/s", flush=True)
    db.close()
    del sdb
