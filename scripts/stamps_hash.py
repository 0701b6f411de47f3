"""Diagnostic: in-kernel phase shares and counters of place_hash64_kernel on a C2-like database spread over a tree of --branches=N
branches (s_memtime stamps, -DRK_STAMPS build; never the product).  --clade: clade-shaped reads (synth.make_clade_db)."""
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
clade = "--clade" in sys.argv
for branches in [int(a.split("=")[1]) for a in sys.argv[1:] if a.startswith("--branches=")] or [7999]:
    n = 2_000_000
    if clade:
        sdb, g = synth.make_clade_db(k=10, n_branches=branches)
        db = ra.PhyloKmerDB.from_synth(sdb)
        pp = ra.PlacementProcess(db)
        seq, offs = synth.make_clade_reads(g, n, 150)
        pk, _, _ = pp.pack_reads_host(seq, offs)
        packed = torch.from_numpy(pk.view(np.int32)).cuda()
    else:
        sdb = synth.make_db(4, 10, branches, 786432, 10_000_000, seed=42)
        db = ra.PhyloKmerDB.from_synth(sdb)
        pp = ra.PlacementProcess(db)
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
    s = a[:4096]
    s = s[s[:, :8].sum(1) > 0]
    print(f"{branches} branches{' (clade-shaped reads)' if clade else ''}: {db.kernel_name()}  {n / (e0.elapsed_time(e1) / 1e3) / 1e6:.1f} Mreads/s (stamped build)")
    placed, handed = s[:, 11].sum(), s[:, 10].sum()
    print(f"  {len(s)} waves, reads placed {int(placed)}, handed over {int(handed)} ({100 * handed / max(1, placed + handed):.2f} %), in doubt (exact ranking) {int(s[:, 13].sum())}")
    t = max(1.0, placed)
    print(f"  per placed read: {s[:, 15].sum() / t:.1f} steps, {s[:, 14].sum() / t:.1f} probe rounds ({s[:, 14].sum() / max(1, s[:, 15].sum()):.2f} a step), {s[:, 12].sum() / t:.2f} steps applied unit by unit")
    tot = s[:, :8].sum()
    for i, nm in enumerate(["read setup + table reset", "probe", "emit", "accumulate", "table scan", "rounds", "exact ranking", "weigh + store"]):
        print(f"    {nm:28s} {100 * s[:, i].sum() / tot:5.1f} %  {s[:, i].sum() / t:9.0f} cycles per placed read")
    db.close()
