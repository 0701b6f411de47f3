"""Developer experiment: would branch-range passes beat the dense score vector on mid-size trees?
Times one pass (the DB restricted to one range of 1000 branches) against the dense run; P passes + a merge would cost ~P x that."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rappas_amd as ra
from rappas_amd import synth

n = 2_000_000
def rate(args):
    db = ra.PhyloKmerDB(*args)
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
    name = db.kernel_name()
    db.close()
    return dt, name

for nb in [int(x) for x in sys.argv[1:]] or [3999, 7999, 12001]:
    sdb = synth.make_db(4, 10, nb, 786432, 10_000_000, seed=42)
    dense, name = rate((4, 10, nb, sdb.thr_log10, sdb.thr, sdb.key_codes, sdb.row_offsets, sdb.branch_ids, sdb.scores))
    RS = 1000
    P = (nb + RS - 1) // RS
    lens = np.diff(sdb.row_offsets.astype(np.int64))
    rowid = np.repeat(np.arange(len(lens)), lens)
    tot = 0.0
    for p in (0, P // 2):
        m = (sdb.branch_ids >= p * RS) & (sdb.branch_ids < (p + 1) * RS)
        cnt = np.bincount(rowid[m], minlength=len(lens))
        keep = cnt > 0
        off = np.zeros(int(keep.sum()) + 1, np.uint64); off[1:] = np.cumsum(cnt[keep])
        t, nm = rate((4, 10, RS, sdb.thr_log10, sdb.thr, sdb.key_codes[keep], off, (sdb.branch_ids[m] - p * RS).astype(np.uint16), sdb.scores[m]))
        tot += t
    per_pass = tot / 2
    print(f"n_branches={nb}: dense {dense*1e3:.1f} ms ({n/dense/1e6:.0f} Mreads/s); one range pass {per_pass*1e3:.1f} ms -> {P} passes {P*per_pass*1e3:.1f} ms "
          f"({n/(P*per_pass)/1e6:.0f} Mreads/s before the merge)", flush=True)
