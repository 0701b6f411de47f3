import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import rappas_amd as ra
from rappas_amd import synth
k, nb, n = 10, 999, 2_000_000
sdb, g = synth.make_clade_db(k=k, n_branches=nb)
db = ra.PhyloKmerDB.from_synth(sdb)
pp = ra.PlacementProcess(db)
rng = np.random.default_rng(6)
starts = rng.integers(0, len(g) - 150, size=n)
for name, st in (("batch order", starts), ("sorted by genome position", np.sort(starts)), ("sorted in blocks of 4096", np.concatenate([np.sort(c) for c in np.array_split(starts, n // 4096)]))):
    idx = (st[:, None] + np.arange(150)[None, :]).reshape(-1)
    seq = np.ascontiguousarray(synth.DNA_LETTERS[g[idx].astype(np.int64)])
    offs = np.arange(n + 1, dtype=np.uint64) * np.uint64(150)
    packed, _, _ = pp.pack_reads_host(seq, offs)
    pk = torch.from_numpy(packed.view(np.int32)).cuda()
    out = pp.place_packed(pk, fixed_len=150); torch.cuda.synchronize()
    t = time.time()
    for _ in range(5):
        pp.place_packed(pk, fixed_len=150, out=out)
    torch.cuda.synchronize()
    print(f"C2-shaped clade DB, reads in {name}: {n / ((time.time() - t) / 5) / 1e6:.1f} Mreads/s", flush=True)
