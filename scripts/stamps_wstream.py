"""Diagnostic: in-kernel phase shares and counters of place_packed16s_kernel (+ the place_packed16w_kernel launch behind it) on a
C2-like database spread over a tree of --branches=N branches (s_memtime stamps, -DRK_STAMPS build; never the product)."""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
so = os.environ.get("RK_STAMPS_LIB") or os.path.join(ROOT, "rappas_amd", "variants", "librk_stamps.so")
if not os.path.exists(so) or "--rebuild" in sys.argv:
    os.makedirs(os.path.dirname(so), exist_ok=True)
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-DRK_DEV_KNOBS", "-DRK_STAMPS",
                    "-o", so, os.path.join(ROOT, "rappas_amd/csrc/rk_engine.hip"), os.path.join(ROOT, "rappas_amd/csrc/rk_pack_host.cpp")], check=True)
os.environ["RK_LIB"] = so
import numpy as np, torch
import rappas_amd as ra
from rappas_amd import synth
for branches in [int(a.split("=")[1]) for a in sys.argv[1:] if a.startswith("--branches=")] or [3999]:
    sdb = synth.make_db(4, 10, branches, 786432, 10_000_000, seed=42)
    db = ra.PhyloKmerDB.from_synth(sdb)
    pp = ra.PlacementProcess(db)
    n = 2_000_000
    wpr = db.packed_words(150)
    packed = torch.randint(-2**31, 2**31, (n, wpr), dtype=torch.int64, device="cuda").to(torch.int32)
    packed[:, wpr - 1] &= (1 << (300 - 32 * (wpr - 1))) - 1
    pp.place_packed(packed, fixed_len=150); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); pp.place_packed(packed, fixed_len=150); e1.record(); torch.cuda.synchronize()
    lib = ra._lib.load()
    buf = (C.c_ulonglong * (8192 * 16))()
    lib.rk_debug_read_stamps.argtypes = [C.c_void_p, C.c_int]
    assert lib.rk_debug_read_stamps(buf, 8192) == 0
    a = np.frombuffer(buf, dtype=np.uint64).reshape(8192, 16).astype(np.float64)
    s, w = a[:4096], a[4096:]
    s = s[s[:, :8].sum(1) > 0]; w = w[w[:, :8].sum(1) > 0]
    print(f"{branches} branches: {db.kernel_name()}  {n / (e0.elapsed_time(e1) / 1e3) / 1e6:.1f} Mreads/s (stamped build)")
    tiles = s[:, 11].sum() + s[:, 10].sum()
    print(f"  sorted-stream kernel: {len(s)} waves, tiles placed {int(s[:, 11].sum())}, handed over {int(s[:, 10].sum())} ({100 * s[:, 10].sum() / max(1, tiles):.1f} %), exact redo {int(s[:, 15].sum())}")
    if s[:, 11].sum():
        t = s[:, 11].sum()
        print(f"  per placed tile: list {s[:, 12].sum() / t:.1f} steps; tiles in doubt because a stream dropped a candidate {int(s[:, 13].sum())} (the rest of the redone ones: ties)")
        tot = s[:, :10].sum()
        for i, nm in enumerate(["tile setup", "emit: S and bitmap reset (the rest of probe + count + sort)", "touched-slot select at window ends", "stream steps", "emit: probe", "rounds", "exact redo (rest in 8/9)", "weigh + store", "emit: counts, segments, fillers", "emit: places + items"]):
            if s[:, i].sum(): print(f"    {nm:40s} {100 * s[:, i].sum() / tot:5.1f} %  {s[:, i].sum() / t:9.0f} cycles per placed tile")
    if len(w): print(f"  place_packed16w_kernel behind it: {len(w)} waves, {w[:, :8].sum() / max(1.0, s[:, :8].sum()) * 100:.1f} % of the first kernel's wave cycles")
    db.close()
