"""Device-side rate of the ambiguity (ASCII) path: reads with a 1 % IUPAC mix, packed + flagged on the device."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rappas_amd as ra
from rappas_amd import synth
sdb = synth.make_config_db("C2")
db = ra.PhyloKmerDB.from_synth(sdb)
pp = ra.PlacementProcess(db)
n = 2_000_000
for rate in (0.0, 0.001, 0.01):
    seq, off = synth.make_reads(4, n, 150, seed=1, amb_rate=rate)
    d_seq = torch.from_numpy(seq).cuda(); d_off = torch.from_numpy(off.view(np.int64)).cuda()
    packed, lens, flags = pp.pack_reads(d_seq, d_off, 150)
    out = pp.place_packed(packed, lens=lens, flags_in=flags, seq_ascii=d_seq, seq_off=d_off)
    torch.cuda.synchronize()
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    e0.record(); packed, lens, flags = pp.pack_reads(d_seq, d_off, 150); e1.record()
    out = pp.place_packed(packed, lens=lens, flags_in=flags, seq_ascii=d_seq, seq_off=d_off); e2.record()
    torch.cuda.synchronize()
    namb = int(((flags & 8) != 0).sum().item())
    print(f"amb_rate={rate}: {namb} ambiguous reads of {n}; pack {e0.elapsed_time(e1):.2f} ms, place(packed+ascii kernels) {e1.elapsed_time(e2):.2f} ms "
          f"-> {n / (e0.elapsed_time(e2) * 1e-3) / 1e6:.1f} Mreads/s incl. pack", flush=True)
