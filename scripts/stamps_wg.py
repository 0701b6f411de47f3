"""Diagnostic: in-kernel phase shares of place_wg_kernel (wave 0 and the last wave of the first 512 workgroups) on a tree of
--branches=N branches with rows of --row=M entries (a quarter of the 9-mers present; -DRK_STAMPS build, never the product)."""
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
arg = lambda n, d: [int(a.split("=")[1]) for a in sys.argv[1:] if a.startswith(f"--{n}=")] or d
for nb in arg("branches", [15999]):
    for row in arg("row", [400, 1000]):
        sdb = synth.make_db(4, 9, nb, 65536, 65536 * row, seed=42)
        db = ra.PhyloKmerDB.from_synth(sdb)
        pp = ra.PlacementProcess(db)
        n = 300_000
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
        a = np.frombuffer(buf, dtype=np.uint64).reshape(8192, 16).astype(np.float64)[:1024]
        print(f"{nb} branches, rows of {row}: {db.kernel_name()[:90]}  {n / (e0.elapsed_time(e1) / 1e3) / 1e6:.1f} Mreads/s (stamped build)")
        for w, nm in ((0, "wave 0"), (1, "last wave")):
            s = a[w::2]
            s = s[s[:, 11] > 0]
            t, tot = s[:, 11].sum(), s[:, :9].sum()
            print(f"  {nm}: segments whose stream heads were in doubt {s[:, 12].sum() / t:.3f} a read, of those through the K-pass fallback {s[:, 13].sum() / t:.3f}")
            print(f"  {nm}: {tot / t:9.0f} cycles per read: " + "  ".join(f"{nmx} {s[:, i].sum() / t:6.0f}" for i, nmx in enumerate(
                ["probe", "compact", "accumulate", "slices", "select-1", "barrier", "weigh+store", "wait(sel-2)", "select-2"])))
        db.close()
