"""Developer measurement: placement rate when a read's best branches are NEIGHBOURS (a clade), the usual shape of real placements.

The C2-like synthetic database scatters a k-mer's row uniformly over the tree, so a read's K best branches fall on random slots.
Here the keys are the k-mers of a random genome and the rows of one 500-bp stretch cover the same few dozen branches; reads are
substrings, so their rows pile up on one neighbourhood and the K best branches are adjacent ids -- the case
in which a select that keeps "the three best of four consecutive slots" per lane has to fall back to its exact path."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rappas_amd as ra
from rappas_amd import synth

k = 10
sizes = [int(x) for x in sys.argv[1:]] or [999, 3999, 7999]
n = 2_000_000
for nb in sizes:
    sdb, g = synth.make_clade_db(k=k, n_branches=nb)
    db = ra.PhyloKmerDB.from_synth(sdb)
    pp = ra.PlacementProcess(db)
    seq, offs = synth.make_clade_reads(g, n, 150)  # reads = genome substrings, packed on the host with the engine's packer
    packed, lens_r, flags = pp.pack_reads_host(seq, offs)
    pk = torch.from_numpy(packed.view(np.int32)).cuda()
    out = pp.place_packed(pk, fixed_len=150); torch.cuda.synchronize()
    t = time.time()
    for _ in range(3):
        pp.place_packed(pk, fixed_len=150, out=out)
    torch.cuda.synchronize()
    dt = (time.time() - t) / 3
    nr = out["n_rows"][:2000].cpu().numpy().astype(np.int64)
    br = out["branch"][:2000].cpu().numpy().astype(np.int64)
    kept = np.arange(br.shape[1])[None, :] < nr[:, None]
    spread = float(np.median((np.where(kept, br, -1).max(1) - np.where(kept, br, 1 << 20).min(1))[nr >= 2])) if (nr >= 2).any() else -1
    print(f"clade n_branches={nb:6d}: {n / dt / 1e6:8.1f} Mreads/s   median id spread of the kept branches = {spread:.0f}   [{db.kernel_name()[:40]}]", flush=True)
    db.close()
