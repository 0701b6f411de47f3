"""Load the hand-derived golden vectors of tests/golden/ (see make_golden.py)."""
import glob
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
FLAG_BITS = {"placed": 1, "bad_char": 2, "too_short": 4, "ambiguous": 8, "below_nsbound": 16}
AMB = {"skip": 0, "mean": 1, "max": 2}


def cases():
    return sorted(glob.glob(os.path.join(HERE, "golden", "*.json")))


def load(path):
    with open(path) as f:
        g = json.load(f)
    codes = np.array([r["code"] for r in g["rows"]], np.uint64)
    off, br, sc = [0], [], []
    for r in g["rows"]:
        for b, vb in r["entries"]:
            br.append(b)
            sc.append(vb)
        off.append(len(br))
    g["csr"] = (codes, np.array(off, np.uint64), np.array(br, np.uint16), np.array(sc, np.uint32).view(np.float32))
    g["T"] = np.array([g["T_bits"]], np.uint32).view(np.float32)[0]
    g["P"] = np.array([g["P_bits"]], np.uint32).view(np.float32)[0]
    return g


def reads_of(run):
    seqs = [e["read"].encode() for e in run["expected"]]
    off = np.zeros(len(seqs) + 1, np.uint64)
    off[1:] = np.cumsum([len(s) for s in seqs])
    seq = np.frombuffer(b"".join(seqs), np.uint8) if off[-1] else np.zeros(0, np.uint8)
    return seq, off


def expected_arrays(run, K):
    exp = run["expected"]
    n = len(exp)
    n_rows = np.zeros(n, np.uint8)
    branch = np.full((n, K), 0xFFFF, np.uint16)
    score = np.full((n, K), 0xFF800000, np.uint32)  # -inf
    lwr = np.zeros((n, K), np.float64)
    flags = np.zeros(n, np.uint32)
    for i, e in enumerate(exp):
        n_rows[i] = len(e["rows"])
        for j, (b, sb, w) in enumerate(e["rows"]):
            branch[i, j], score[i, j], lwr[i, j] = b, sb, float(w)
        for f in e["flags"]:
            flags[i] |= FLAG_BITS[f]
    return n_rows, branch, score.view(np.float32), lwr, flags


def has_tie(e):
    """exact score tie among the touched branches of a golden read (tie order is layout-dependent in the reference)."""
    vals = sorted(e["S"].values())
    return any(a == b for a, b in zip(vals, vals[1:]))


def build_cases():
    return sorted(glob.glob(os.path.join(HERE, "golden", "build", "*.json")))


def load_build(path):
    """tests/golden/build/*.json (make_golden_build.py) -> kwargs for the builders + expected CSR."""
    with open(path) as f:
        g = json.load(f)
    states = np.array(g["states"], np.uint8)
    pp = np.array(g["pp_bits"], np.uint32).view(np.float32)
    kw = {}
    if g["gaps"] is not None:
        off = np.zeros(len(g["gaps"]) + 1, np.uint32)
        off[1:] = np.cumsum([len(x) for x in g["gaps"]])
        kw = dict(gap_off=off, gap_len=np.array([v for x in g["gaps"] for v in x], np.int32), limit_to_1_jump=g["limit_to_1_jump"])
    T = np.array([g["T_bits"]], np.uint32).view(np.float32)[0]
    e = g["expected"]
    exp = dict(key_codes=np.array(e["key_codes"], np.uint64), row_offsets=np.array(e["row_offsets"], np.uint64),
               branch_ids=np.array(e["branch_ids"], np.uint16), score_bits=np.array(e["score_bits"], np.uint32),
               tuples=e["tuples"], visits=e["visits"])
    return (g["alphabet"], g["k"], states, pp, np.array(g["node_branch"], np.uint16), T), kw, exp
