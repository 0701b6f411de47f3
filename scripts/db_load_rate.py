"""Developer measurement: how long the native driver takes to have C2's database on the GPU from a --jsondb dump and from a --dbimage
file, and its FASTA -> jplace rate through the classic (round 3) and the all-threads (round 4) host path.  -> profiles/r04_db_load_rate.txt"""
import json, os, shutil, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import rappas_amd as ra
from rappas_amd import build, hostio, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
exe = build.build_host_tools()
sdb = synth.make_config_db("C2")
nwk = synth.make_newick(sdb.n_branches, seed=3)
d = tempfile.mkdtemp(prefix="rk_dbl_", dir="/dev/shm")
try:
    t = time.time(); open(d + "/db.json", "w").write(hostio.dump_jsondb(sdb, nwk)); print(f"--jsondb dump written: {os.path.getsize(d + '/db.json') / 1e6:.0f} MB ({time.time() - t:.0f} s in Python)", flush=True)
    db = ra.PhyloKmerDB.from_synth(sdb)
    db.save(d + "/db.rkimg", user=hostio.tree_to_blob(hostio.parse_newick(nwk)))
    print(f"image file: {os.path.getsize(d + '/db.rkimg') / 1e6:.0f} MB", flush=True)
    seq, off = synth.make_reads(4, n, 150, seed=1)
    with open(d + "/q.fasta", "wb") as f:
        s = np.asarray(seq).reshape(n, 150)
        for i in range(0, n, 100000):
            f.write(b"".join(b">r%07d\n%s\n" % (j, s[j].tobytes()) for j in range(i, min(n, i + 100000))))
    for name, args in [("--jsondb, classic host path (round 3)", ["--jsondb", d + "/db.json", "--classic-io"]), ("--jsondb, all-threads host path", ["--jsondb", d + "/db.json"]),
                       ("--dbimage, classic host path", ["--dbimage", d + "/db.rkimg", "--classic-io"]), ("--dbimage, all-threads host path (round 4)", ["--dbimage", d + "/db.rkimg"])]:
        best = None
        for _ in range(2):
            t = time.time()
            r = subprocess.run([exe] + args + ["--fasta", d + "/q.fasta", "--out", d + "/q.jplace", "--keep-at-most", "7", "--threads", "32", "--timing"], capture_output=True, text=True, timeout=900)
            wall = time.time() - t
            if r.returncode != 0:
                print(name, "FAILED", r.stderr[-300:]); break
            p = json.loads(r.stdout.strip().splitlines()[-1])
            p["wall_s"] = wall
            if best is None or p["fasta_to_jplace_s"] < best["fasta_to_jplace_s"]:
                best = p
        if best:
            print(f"{name:45s}: database ready in {best['db_s']:6.2f} s; {n} reads FASTA -> jplace in {best['fasta_to_jplace_s']:6.3f} s = {n / best['fasta_to_jplace_s'] / 1e6:5.2f} Mreads/s; process wall {best['wall_s']:.2f} s", flush=True)
finally:
    shutil.rmtree(d, ignore_errors=True)
