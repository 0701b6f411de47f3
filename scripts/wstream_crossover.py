"""Developer measurement: place_packed16w_kernel against place_packed16s_kernel on mid-size trees (C2-like DB, 150 bp reads), to place
RK_WSTREAM_MIN_BRANCHES.  Needs the developer build (RK_WSTREAM_ALWAYS / RK_NO_WSTREAM are its knobs); one process per point."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--one":
    sys.path.insert(0, ROOT)
    import time, torch
    import rappas_amd as ra
    from rappas_amd import synth
    import numpy as np
    nb, n = int(sys.argv[2]), 2_000_000
    aa = len(sys.argv) > 3 and sys.argv[3] == "aa"  # amino acids k=5, 100-residue reads (C4-like rows) instead of DNA k=10, 150 bp
    db = ra.PhyloKmerDB.from_synth(synth.make_db(20, 5, nb, 786432, 10_000_000, seed=42) if aa else synth.make_db(4, 10, nb, 786432, 10_000_000, seed=42))
    pp = ra.PlacementProcess(db)
    rlen = 100 if aa else 150
    if aa:  # digits have to stay below 20: real reads, packed on the device
        seq, off = synth.make_reads(20, n, rlen, seed=1)
        packed, _, _ = pp.pack_reads(torch.from_numpy(seq).cuda(), torch.from_numpy(off.view(np.int64)).cuda(), rlen)
    else:
        wpr = db.packed_words(rlen)
        packed = torch.randint(-2**31, 2**31, (n, wpr), dtype=torch.int64, device="cuda").to(torch.int32)
        packed[:, wpr - 1] &= (1 << (2 * rlen - 32 * (wpr - 1))) - 1
    out = pp.place_packed(packed, fixed_len=rlen); torch.cuda.synchronize()
    t = time.time()
    for _ in range(3):
        pp.place_packed(packed, fixed_len=rlen, out=out)
    torch.cuda.synchronize()
    print(f"{n * 3 / (time.time() - t) / 1e6:8.1f} Mreads/s  [{db.kernel_name()[:70]}]")
    sys.exit(0)
aa = ["aa"] if "aa" in sys.argv[1:] else []
for nb in [int(x) for x in sys.argv[1:] if x != "aa"] or [2999, 3999, 4999, 5999, 6999, 7999]:
    for knob in ("RK_NO_WSTREAM", "RK_WSTREAM_ALWAYS"):
        env = dict(os.environ, RK_LIB=os.path.join(ROOT, "rappas_amd", "librappas_place_dev.so"))
        env[knob] = "1"
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--one", str(nb)] + aa, env=env, capture_output=True, text=True)
        print(f"n_branches={nb:6d} {knob:18s} {(r.stdout.strip() or r.stderr.strip()[-200:])}", flush=True)
