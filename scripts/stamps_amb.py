"""Diagnostic: in-kernel phase shares of place_ascii_kernel (s_memtime stamps, -DRK_STAMPS build; never the product)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["RK_LIB"] = os.environ.get("RK_STAMPS_LIB") or os.path.join(ROOT, "rappas_amd", "variants", "librk_stamps.so")
import numpy as np, torch
import rappas_amd as ra
from rappas_amd import synth
rate = float(sys.argv[1]) if len(sys.argv) > 1 else 0.001
sdb = synth.make_config_db("C2")
db = ra.PhyloKmerDB.from_synth(sdb)
pp = ra.PlacementProcess(db)
n = 1_000_000
seq, off = synth.make_reads(4, n, 150, seed=1, amb_rate=rate)
d_seq = torch.from_numpy(seq).cuda(); d_off = torch.from_numpy(off.view(np.int64)).cuda()
packed, lens, flags = pp.pack_reads(d_seq, d_off, 150)
keep = (flags & 8) != 0  # only the ambiguous reads: the packed kernel then has nothing to stamp over
idx = torch.nonzero(keep).flatten()
for _ in range(2):
    out = pp.place_packed(packed, lens=lens, flags_in=flags, seq_ascii=d_seq, seq_off=d_off)
    torch.cuda.synchronize()
lib = ra._lib.load()
nw = 4096
buf = (C.c_ulonglong * (nw * 16))()
lib.rk_debug_read_stamps.argtypes = [C.c_void_p, C.c_int]
assert lib.rk_debug_read_stamps(buf, nw) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(nw, 16).astype(np.float64)
a = a[a[:, 10] > 0]
tot = a[:, :10].sum(1)
reads = a[:, 10]
names = ["flag inspection / between reads", "decode + probes", "list building", "flush (accumulate)", "ambiguous positions", "select", "weigh + store"]
print(f"{int(keep.sum())} ambiguous reads of {n}; {len(a)} waves; median cycles per read {np.median(tot / reads):.0f}")
for i, nm in enumerate(names):
    print(f"  {nm:34s} {100 * np.median(a[:, i] / tot):5.1f} %   {np.median(a[:, i] / reads):8.0f} cycles/read")
print(f"  (select: scan {np.median(a[:, 8] / reads):.0f}, rounds {np.median(a[:, 9] / reads):.0f})")
