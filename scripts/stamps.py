"""Diagnostic: in-kernel phase shares of place_packed_kernel (s_memtime stamps, -DRK_STAMPS build; never the product)."""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
so = os.path.join(ROOT, "rappas_amd", "variants", "librk_stamps.so")
extra = sys.argv[1:]
flags = [a for a in extra if a.startswith("-D")]
table = next((a.split("=")[1] for a in extra if a.startswith("--table=")), "auto")
branches = int(next((a.split("=")[1] for a in extra if a.startswith("--branches=")), "0"))  # mid-size tree (windowed kernel) instead of C2
if flags or not os.path.exists(so):  # (prebuild it on the build host into rappas_amd/variants/ to save GPU-box time)
    os.makedirs(os.path.dirname(so), exist_ok=True)
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-DRK_DEV_KNOBS", "-DRK_STAMPS"] + flags +
                   ["-o", so, os.path.join(ROOT, "rappas_amd/csrc/rk_engine.hip"), os.path.join(ROOT, "rappas_amd/csrc/rk_pack_host.cpp")], check=True)
os.environ["RK_LIB"] = so
import numpy as np, torch
import rappas_amd as ra
from rappas_amd import synth
cfg = next((a.split("=")[1] for a in extra if a.startswith("--config=")), "C2")  # C2 or C4
sdb = synth.make_db(4, 10, branches, 786432, 10_000_000, seed=42) if branches else synth.make_config_db(cfg)
rlen = synth.CONFIGS[cfg][5] if not branches else 150
bits = 2 if sdb.alphabet == 4 else 5
mode = {"auto": ra.RK_TABLE_AUTO, "direct": ra.RK_TABLE_DIRECT, "direct8": ra.RK_TABLE_DIRECT8, "hash": ra.RK_TABLE_HASH}[table]
db = ra.PhyloKmerDB.from_synth(sdb, table_mode=mode)
pp = ra.PlacementProcess(db)
n = 4_000_000
if bits == 2:
    wpr = db.packed_words(rlen)
    packed = torch.randint(-2**31, 2**31, (n, wpr), dtype=torch.int64, device="cuda").to(torch.int32)
    packed[:, wpr - 1] &= (1 << (2 * rlen - 32 * (wpr - 1))) - 1
else:  # amino acids: digits must stay below 20 -- pack real reads on the device
    seq, off = synth.make_reads(20, n, rlen, seed=1)
    packed, _, _ = pp.pack_reads(torch.from_numpy(seq).cuda(), torch.from_numpy(off.view(np.int64)).cuda(), rlen)
pp.place_packed(packed, fixed_len=rlen); torch.cuda.synchronize()
pp.place_packed(packed, fixed_len=rlen); torch.cuda.synchronize()
lib = ra._lib.load()
nw = 2048
buf = (C.c_ulonglong * (nw * 16))()
lib.rk_debug_read_stamps.argtypes = [C.c_void_p, C.c_int]
assert lib.rk_debug_read_stamps(buf, nw) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(nw, 16).astype(np.float64)
tot = a.sum(1)
names = ["tile setup", "probe (codes+gathers)", "scan+emit items", "pre-accumulate fence", "accumulate", "select (rest: reset)", "weigh+store", "-",
         "select: scan", "select: rounds"]
windowed = "packed16w" in db.kernel_name()
if windowed:
    names = ["tile setup", "probe + emit", "window compaction", "window accumulate", "window scan + reset", "rounds", "redo (doubt)", "weigh+store"]
a = a[tot > 0]
tot = tot[tot > 0]
nw = len(tot)
print(db.kernel_name())
print("median wave cycles:", np.median(tot), " per tile:", np.median(tot) / (n / 4 / nw))
if windowed:
    tot = a[:, :10].sum(1)
for i, nm in enumerate(names):
    print(f"  {nm:24s} {100 * np.median(a[:, i] / tot):5.1f} %   {np.median(a[:, i]) / (n / 4 / nw):8.0f} cycles/tile")
if windowed:
    tiles = n / 4 / nw
    print(f"  accumulate calls/tile {np.median(a[:, 11]) / tiles:.2f}, steps/call {np.median(a[:, 10] / a[:, 11]):.1f}, "
          f"cycles/call {np.median(a[:, 3] / a[:, 11]):.0f}, cycles/step {np.median(a[:, 3] / a[:, 10]):.1f}")
