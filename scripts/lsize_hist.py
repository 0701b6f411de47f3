"""Developer measurement (CPU only): how many DISTINCT branches does a read touch -- the length of the reference's touched list
L (PlacementProcess.java:726-729) -- on the large short-row trees, uniform and clade-shaped reads?  Prices a per-read hash
accumulator in the LDS: its table has to hold |L| keys, whatever the tree's size.

Counts come from the CSR arrays with numpy; a sample of reads is cross-checked against the oracle's ro_score_vector.
usage: python scripts/lsize_hist.py [n_reads]   ->  profiles/r04_lsize_hist.txt
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from rappas_amd import synth
from oracle import oracle as O


def lsize(sdb, seq, off, check=8):
    k, n = sdb.k, len(off) - 1
    R = int(off[1] - off[0])
    st = np.zeros(256, np.uint64)
    for i, c in enumerate(b"ATCG"):
        st[c] = i
    s = st[seq].reshape(n, R)
    Q = R - k + 1
    codes = np.zeros((n, Q), np.uint64)
    for i in range(k):
        codes += s[:, i:i + Q] << np.uint64(2 * i)
    dense = np.full(4 ** k, -1, np.int64)
    dense[sdb.key_codes.astype(np.int64)] = np.arange(sdb.n_keys)
    ro = sdb.row_offsets.astype(np.int64)
    sizes = np.zeros(n, np.int64)
    ents = np.zeros(n, np.int64)
    units = np.zeros(n, np.int64)
    for r in range(n):
        rows = dense[codes[r].astype(np.int64)]
        rows = rows[rows >= 0]
        lens = ro[rows + 1] - ro[rows]
        ents[r] = lens.sum()
        units[r] = ((lens + 15) // 16).sum()
        br = np.concatenate([sdb.branch_ids[ro[x]:ro[x + 1]] for x in rows]) if len(rows) else np.zeros(0, np.uint16)
        sizes[r] = len(np.unique(br))
    if check:
        odb = O.OracleDB.from_synth(sdb)
        for r in range(min(check, n)):
            _, L, _ = odb.score_vector(bytes(seq[int(off[r]):int(off[r + 1])]))
            assert len(L) == sizes[r], (r, len(L), sizes[r])
        odb.close()
    return sizes, ents, units


def line(tag, a):
    q = np.percentile(a, [50, 90, 95, 99, 99.9])
    return f"{tag:<34s} mean {a.mean():7.1f}  p50 {q[0]:6.0f}  p90 {q[1]:6.0f}  p95 {q[2]:6.0f}  p99 {q[3]:6.0f}  p99.9 {q[4]:6.0f}  max {a.max():6d}"


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
    out = [f"# |L| = distinct branches a read touches (oracle-checked on 8 reads per line), entries H and 128-byte row units per read; {n} reads of 150 bp per line",
           "# scripts/lsize_hist.py; uniform = bench.py --config T8k/T20k/T64k (C2's rows on the tree named, reads seed 1); clade = synth.make_clade_db / make_clade_reads"]
    for name, nb in (("T8k", 7999), ("T20k", 19999), ("T64k", 65535)):
        sdb = synth.make_config_db(name)
        seq, off = synth.make_reads(4, n, 150, seed=1)
        sz, en, un = lsize(sdb, seq, off)
        out.append(line(f"{name} uniform |L|", sz))
        out.append(line(f"{name} uniform entries", en))
        out.append(line(f"{name} uniform row units", un))
        for cap in (1024, 1536, 2048, 3072):
            out.append(f"    reads with |L| > {cap}: {100.0 * (sz > cap).mean():.2f} %")
        cdb, g = synth.make_clade_db(k=10, n_branches=nb)
        seq, off = synth.make_clade_reads(g, n, 150)
        sz, en, un = lsize(cdb, seq, off)
        out.append(line(f"{name} clade   |L|", sz))
        out.append(line(f"{name} clade   entries", en))
        out.append(line(f"{name} clade   row units", un))
        print("\n".join(out[-9:]), flush=True)
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r04_lsize_hist.txt")
    with open(path, "w") as f:
        f.write("\n".join(out) + "\n")


if __name__ == "__main__":
    main()
