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
rng = np.random.default_rng(5)
glen = 700_000
g = rng.integers(0, 4, size=glen).astype(np.uint64)
codes = np.zeros(glen - k + 1, dtype=np.uint64)
for i in range(k):
    codes += g[i:glen - k + 1 + i] << np.uint64(2 * i)
key_codes, first = np.unique(codes, return_index=True)
order = rng.permutation(len(key_codes))
key_codes, pos = key_codes[order], first[order]
for nb in sizes:
    lens = np.minimum(rng.geometric(1.0 / 12.7, size=len(key_codes)), nb - 1).astype(np.int64)
    off = np.zeros(len(key_codes) + 1, dtype=np.uint64)
    np.cumsum(lens, out=off[1:])
    hi = np.maximum(1, nb - lens)
    block = pos // 500  # 500-bp stretches of the genome share a neighbourhood of the tree
    b0 = np.clip(1 + block * np.maximum(1, hi - 40) // (glen // 500 + 1) + rng.integers(-6, 7, size=len(pos)), 1, hi)
    total = int(off[-1])
    within = np.arange(total, dtype=np.int64) - np.repeat(off[:-1].astype(np.int64), lens)
    branch = (np.repeat(b0, lens) + within).astype(np.uint16)
    thr, thr_log10 = synth.thresholds(1.5, 4, k)
    scores = (thr_log10 * rng.random(total, dtype=np.float32)).astype(np.float32)
    sdb = synth.SynthDB(4, k, nb, thr, thr_log10, key_codes, off, branch, scores, 5)
    db = ra.PhyloKmerDB.from_synth(sdb)
    pp = ra.PlacementProcess(db)
    starts = rng.integers(0, glen - 150, size=n)
    # reads = genome substrings, packed on the host with the engine's packer
    idx = (starts[:, None] + np.arange(150)[None, :]).reshape(-1)
    seq = np.frombuffer(b"ATCG", dtype=np.uint8)[g[idx].astype(np.int64)]
    offs = (np.arange(n + 1, dtype=np.uint64) * np.uint64(150))
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
