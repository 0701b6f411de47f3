"""Developer measurement: place_hash64_kernel against place_packed16s_kernel on large short-row trees (C2-like DB, 150 bp reads,
uniform and clade-shaped), to place RK_HASH_MIN_BRANCHES.  Needs the developer build (RK_HASH_ALWAYS / RK_NO_HASH are its knobs);
one process per point.  -> profiles/r04_hash_crossover.txt"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--one":
    sys.path.insert(0, ROOT)
    import time, torch
    import rappas_amd as ra
    from rappas_amd import synth
    import numpy as np
    nb, shape, n = int(sys.argv[2]), sys.argv[3], 2_000_000
    if shape == "clade":
        sdb, g = synth.make_clade_db(k=10, n_branches=nb)
        db = ra.PhyloKmerDB.from_synth(sdb)
        pp = ra.PlacementProcess(db)
        seq, offs = synth.make_clade_reads(g, n, 150)
        packed = torch.from_numpy(pp.pack_reads_host(seq, offs)[0].view(np.int32)).cuda()
    elif shape == "aa-clade":  # amino acids k = 5, 100-residue reads cut from the "genome" whose k-mers make up the database
        sdb, g = synth.make_clade_db(k=5, n_branches=nb, alphabet=20)
        db = ra.PhyloKmerDB.from_synth(sdb)
        pp = ra.PlacementProcess(db)
        seq, offs = synth.make_clade_reads(g, n, 100, alphabet=20)
        packed, _, _ = pp.pack_reads(torch.from_numpy(seq).cuda(), torch.from_numpy(offs.view(np.int64)).cuda(), 100)
    elif shape == "aa":  # amino acids k = 5, 100-residue reads on C4-like rows
        db = ra.PhyloKmerDB.from_synth(synth.make_db(20, 5, nb, 786432, 10_000_000, seed=42))
        pp = ra.PlacementProcess(db)
        seq, off = synth.make_reads(20, n, 100, seed=1)
        packed, _, _ = pp.pack_reads(torch.from_numpy(seq).cuda(), torch.from_numpy(off.view(np.int64)).cuda(), 100)
    else:
        db = ra.PhyloKmerDB.from_synth(synth.make_db(4, 10, nb, 786432, 10_000_000, seed=42))
        pp = ra.PlacementProcess(db)
        wpr = db.packed_words(150)
        packed = torch.randint(-2**31, 2**31, (n, wpr), dtype=torch.int64, device="cuda").to(torch.int32)
        packed[:, wpr - 1] &= (1 << (300 - 32 * (wpr - 1))) - 1
    rl = 100 if shape.startswith("aa") else 150
    out = pp.place_packed(packed, fixed_len=rl); torch.cuda.synchronize()
    t = time.time()
    for _ in range(3):
        pp.place_packed(packed, fixed_len=rl, out=out)
    torch.cuda.synchronize()
    print(f"{n * 3 / (time.time() - t) / 1e6:8.1f} Mreads/s  [{db.kernel_name()[:34]}]")
    sys.exit(0)
shapes = ("aa", "aa-clade") if "aa" in sys.argv[1:] else ("uniform", "clade")
for nb in [int(x) for x in sys.argv[1:] if x != "aa"] or [12001, 19999, 28001, 39999, 65535]:
    for shape in shapes:
        for knob in ("RK_NO_HASH", "RK_HASH_ALWAYS", "(the engine's choice)"):
            env = dict(os.environ, RK_LIB=os.environ.get("RK_VARIANT_LIB") or os.path.join(ROOT, "rappas_amd", "librappas_place_dev.so"))
            if knob.startswith("RK_"):
                env[knob] = "1"
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--one", str(nb), shape], env=env, capture_output=True, text=True)
            print(f"n_branches={nb:6d} {shape:8s} {knob:21s} {(r.stdout.strip() or r.stderr.strip()[-200:])}", flush=True)
