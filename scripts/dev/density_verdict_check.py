"""Developer check (under rocprofv3 --kernel-trace): which instantiation of place_hash64_kernel runs for a batch of random reads and for a
batch of reads all of whose k-mers have a row, on a database that holds a fifth of the k-mer codes (20 001 branches).  argv[1]: sparse | dense"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import rappas_amd as ra
from rappas_amd import synth
sdb, g = synth.make_clade_db(k=10, n_branches=20001, genome_len=200_000, seed=9)
rng = np.random.default_rng(4)
lens = np.diff(sdb.row_offsets.astype(np.int64))
b0 = rng.integers(1, np.maximum(2, sdb.n_branches - lens))
within = np.arange(int(sdb.row_offsets[-1]), dtype=np.int64) - np.repeat(sdb.row_offsets[:-1].astype(np.int64), lens)
sdb = synth.SynthDB(sdb.alphabet, sdb.k, sdb.n_branches, sdb.thr, sdb.thr_log10, sdb.key_codes, sdb.row_offsets, (np.repeat(b0, lens) + within).astype(np.uint16), sdb.scores, sdb.seed)
db = ra.PhyloKmerDB.from_synth(sdb)
pp = ra.PlacementProcess(db)
n = 200_000
seq, off = (synth.make_clade_reads(g, n, 150, seed=3) if sys.argv[1] == "dense" else synth.make_reads(4, n, 150, seed=6))
packed, _, _ = pp.pack_reads(torch.from_numpy(seq).cuda(), torch.from_numpy(off.view(np.int64)).cuda(), 150)
for _ in range(3):
    pp.place_packed(packed, fixed_len=150)
torch.cuda.synchronize()
print(db.kernel_name())
