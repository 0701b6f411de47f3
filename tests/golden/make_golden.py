#!/usr/bin/env python3
"""Authoring script of the hand-derived known-answer vectors in this directory.

The reference ships no tests or fixtures for the placement path and cannot run here, so
these vectors are derived from its source: tests/pyref.py spells out every float32/float64 operation of
PlacementProcess.java:645-1075 in plain Python; this script applies it to small hand-written DBs/reads and stores
inputs + expected outputs as JSON.  The C oracle (oracle/) and the HIP engine are then checked against the JSON.
Scores are stored as float32 bit patterns (exactness), LWRs as repr() of the double.

Run from the repo root:  python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from tests import pyref  # noqa: E402

f32 = np.float32


def bits(x):
    return int(f32(x).view(np.uint32))


def dna_code(s):
    return pyref.kmer_code(4, [pyref.DNA_STATES[c] for c in s])


def aa_code(s):
    return pyref.kmer_code(20, [pyref.AA_ORDER.index(c) for c in s])


def make_case(name, alphabet, k, n_branches, rows, reads, params_list, omega=1.5, note=""):
    P, T = pyref.thresholds(omega, alphabet, k)
    db = dict(alphabet=alphabet, k=k, n_branches=n_branches, T=T, P=P,
              rows={code: [(b, f32(T * f32(u))) for b, u in ents] for code, ents in rows.items()})
    out = dict(name=name, note=note, alphabet=alphabet, k=k, n_branches=n_branches, omega=omega,
               T_bits=bits(T), P_bits=bits(P),
               rows=[dict(code=int(c), entries=[[int(b), bits(v)] for b, v in ents]) for c, ents in db["rows"].items()],
               runs=[])
    for params in params_list:
        exp = []
        for rd in reads:
            r = pyref.place_read(db, rd, **params)
            exp.append(dict(read=rd, flags=sorted(r["flags"]), H=r["H"],
                            rows=[[int(b), bits(s), repr(float(w))] for b, s, w in r["rows"]],
                            L=[int(x) for x in r["L"]],
                            S={str(x): bits(v) for x, v in r["S"].items()}))
        out["runs"].append(dict(params=params, expected=exp))
    with open(os.path.join(HERE, name + ".json"), "w") as f:
        json.dump(out, f, indent=1)
    return out


def main():
    default = dict(keep_at_most=7, keep_factor=0.01, amb_mode="mean")
    # ---- 1. toy DNA DB, k=4, 6 branches: plain / case / U / ambiguity / edge lengths ----
    rows = {
        dna_code("ACGT"): [(1, 0.10), (2, 0.50), (3, 0.90)],
        dna_code("CGTA"): [(2, 0.20), (3, 0.30)],
        dna_code("GTAC"): [(3, 0.05), (1, 0.60), (5, 0.70)],
        dna_code("TACG"): [(4, 0.99)],
        dna_code("AAAA"): [(1, 0.25), (4, 0.26)],
        dna_code("ACGA"): [(2, 0.40)],
        dna_code("ACGC"): [(5, 0.15), (2, 0.35)],
        dna_code("ACGG"): [(1, 0.45)],
        dna_code("TTTT"): [(1, 0.0), (2, 1.0)],
    }
    reads = [
        "ACGTACGTAC",            # repeated hits, per-branch sums in k-mer order
        "acgtacgtac",            # lower case = same states (DNAStatesShifted.java:182-209)
        "ACGUACGUAC",            # U -> T
        "ACGNACGT",              # one N per window: mean/max/skip; alternatives in order A,C,G,T (:92-96)
        "ACGRACGT",              # R = {A,G}
        "ACNNACGTACGT",          # two ambiguities in a window -> skipped but counted in Q
        "AC-TACGTA",             # '-' = four zero alternatives {A,A,A,A} (:57-58)
        "GGGGGGGG",              # no hit -> unplaced
        "ACG",                   # R = k-1 -> Q = 0, unplaced
        "AC",                    # R < k-1 (reference crashes) -> too_short
        "ACGT",                  # R = k, single k-mer
        "ACGTXACGT",             # unsupported character -> bad_char
        "TTTTTTTT",              # v == T (d = 0) and v == 0 entries
        "",                      # empty read
    ]
    make_case("toy_dna_k4", 4, 4, 6, rows, reads,
              [default, dict(default, amb_mode="max"), dict(default, amb_mode="skip"),
               dict(default, keep_at_most=2), dict(default, keep_at_most=1), dict(default, keep_factor=0.9),
               dict(default, keep_factor=0.0), dict(default, ns_bound=-5.0)],
              note="shift==0 side of the -308 rule (short reads, Q*T ~ -10)")

    # ---- 2. long reads: Q*T < -308 -> weightRatioShift = best (PlacementProcess.java:384-390) ----
    unit = "ACGTACGAACGCACGGTACG"
    long_reads = [unit * 12, (unit * 12)[3:], "ACGT" * 50 + "AAAA" * 3, "AAAA" * 60, (unit * 10) + "N" + (unit * 2)]
    make_case("shift_dna_k4", 4, 4, 6, rows, long_reads,
              [default, dict(default, keep_factor=0.5), dict(default, keep_at_most=3), dict(default, amb_mode="max")],
              note="Q*T <= -308: shifted sums; keep-factor cut stops at the first rank below factor*best")

    # ---- 3. more touched branches than keep_at_most, exact ties included ----
    rows3 = {
        dna_code("AAAAA"): [(i, 0.05 * i) for i in range(1, 12)],
        dna_code("AAAAC"): [(i, 0.5) for i in range(3, 9)],
        dna_code("CCCCC"): [(2, 0.3), (9, 0.3), (4, 0.3)],
    }
    reads3 = ["AAAAAC", "AAAAA", "CCCCC", "CCCCCC", "AAAAACCCCC"]
    make_case("topk_dna_k5", 4, 5, 12, rows3, reads3, [default, dict(default, keep_at_most=3), dict(default, keep_factor=0.0)],
              note="12 branches > keep_at_most; CCCCC rows give exact float ties (order = java.util.PriorityQueue on L order)")

    # ---- 4. amino acids, k=3 ----
    rows4 = {
        aa_code("RHK"): [(1, 0.2), (2, 0.4)],
        aa_code("HKD"): [(2, 0.1), (3, 0.7)],
        aa_code("KDE"): [(1, 0.3)],
        aa_code("DKD"): [(3, 0.6)],
        aa_code("NKD"): [(1, 0.8), (3, 0.2)],
        aa_code("VVV"): [(2, 0.5)],
        aa_code("CKD"): [(1, 0.11)],
    }
    reads4 = ["RHKDE", "rhkde", "RHKBKDE", "RHXDE", "VVVV", "RH", "RHKDE*VVV", "RHKUKD", "OOO"]
    make_case("toy_aa_k3", 20, 3, 4, rows4, reads4, [default, dict(default, amb_mode="max"), dict(default, amb_mode="skip")],
              note="AA order RHKDESTNQCGPAILMFWYV; B={D,N}; X/* = all 20; U/O unsupported without convertUO")
    print("golden vectors written to", HERE)


if __name__ == "__main__":
    main()
