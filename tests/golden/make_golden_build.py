#!/usr/bin/env python3
"""Authoring script of the known-answer vectors in tests/golden/build/ (phylo-kmer DB construction).

The reference ships no fixtures for this path and cannot run here, so the vectors are derived from its source:
tests/pyref_build.py replays src/core/algos/WordExplorer_v3.java:98-199 and its callers in plain Python (every float32
operation spelled out); this script applies it to small hand-written posterior tables and stores inputs + expected output as
JSON.  The C oracle (oracle/rappas_build_oracle.c) and the GPU builder (rk_build_db) are then checked against the JSON.
Floats are stored as float32 bit patterns.

Run from the repo root:  python tests/golden/make_golden_build.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from tests import pyref_build as PB  # noqa: E402

f32 = np.float32


def ranked(prob_rows):
    """[[p_A, p_T, p_C, p_G], ...] per site -> (states ranked by descending p (stable), log10 p as float32)"""
    st, pp = [], []
    for row in prob_rows:
        order = sorted(range(len(row)), key=lambda s: -row[s])
        st.append(order)
        pp.append([f32(np.log10(np.float64(f32(row[s])))) for s in order])
    return st, pp


def gap_lists(rows):
    L = len(rows[0])
    out = [[] for _ in range(L)]
    for r in rows:  # Alignment.updateGapIntervals (src/alignement/Alignment.java:232-258)
        first = -1
        for j, c in enumerate(r):
            if c == "-":
                if first == -1:
                    first = j
            elif first != -1:
                if (j - first) not in out[first]:
                    out[first].append(j - first)
                first = -1
    return out


def make(name, alphabet, k, omega, node_probs, node_branch, align_rows=None, limit1=True, note=""):
    ratio = f32(omega) / f32(alphabet)
    T = f32(np.log10(float(f32(float(ratio) ** k))))  # Main_DBBUILD_3.java:165-166
    st, pp = zip(*(ranked(n) for n in node_probs))
    states = np.array(st, np.uint8)
    ppa = np.array(pp, np.float32)
    gaps = gap_lists(align_rows) if align_rows else None
    sink, tuples, visits = PB.build(alphabet, k, states, ppa, node_branch, T, gaps=gaps, limit1=limit1)
    codes, off, br, sc = PB.to_csr(sink)
    doc = dict(name=name, note=note, alphabet=alphabet, k=k, omega=omega, T_bits=int(T.view(np.uint32)),
               states=states.tolist(), pp_bits=ppa.view(np.uint32).tolist(), node_branch=list(map(int, node_branch)),
               gaps=gaps, limit_to_1_jump=limit1,
               expected=dict(key_codes=codes.tolist(), row_offsets=off.tolist(), branch_ids=br.tolist(),
                             score_bits=sc.view(np.uint32).tolist(), tuples=tuples, visits=visits))
    with open(os.path.join(HERE, "build", name + ".json"), "w") as f:
        json.dump(doc, f, indent=1)
    print(name, "keys", len(codes), "entries", len(br), "tuples", tuples, "visits", visits)


if __name__ == "__main__":
    conserved = lambda s, p=0.97: [p if i == s else (1 - p) / 3 for i in range(4)]
    node_a = [conserved(0), conserved(2), [0.5, 0.3, 0.15, 0.05], conserved(3), conserved(1, 0.9), [0.4, 0.4, 0.1, 0.1], conserved(0), conserved(2)]
    node_b = [conserved(0, 0.8), conserved(1), [0.25, 0.25, 0.25, 0.25], conserved(3), conserved(1), conserved(2, 0.6), conserved(0), conserved(3)]
    node_c = [conserved(2), conserved(2), conserved(2), conserved(2), [0.7, 0.1, 0.1, 0.1], conserved(0), conserved(1), conserved(1)]
    make("dna_k3_three_nodes", 4, 3, 1.5, [node_a, node_b, node_c], [4, 4, 9],
         note="two nodes share branch 4 (maximum across nodes), uniform and tied columns (stable rank order), 8 sites")
    make("dna_k4_omega1", 4, 4, 1.0, [node_a, node_b], [0, 1], note="omega=1.0: tighter threshold, most words pruned")
    rows = ["AC-GT-CA", "ACGGTACA", "A--GTTCA"]
    make("dna_k3_gap_one_jump", 4, 3, 1.5, [node_a, node_c], [2, 3], align_rows=rows, limit1=True,
         note="gap jumps, at most one jump per first state (limitTo1Jump, sticky idxOfFirstJump)")
    make("dna_k3_gap_all_jumps", 4, 3, 1.5, [node_a, node_c], [2, 3], align_rows=rows, limit1=False,
         note="gap jumps, all combinations")
    aa = []
    rng = [0.31, 0.17, 0.11, 0.09, 0.07, 0.05, 0.04, 0.03, 0.03, 0.02, 0.02, 0.01, 0.01, 0.01, 0.01, 0.005, 0.005, 0.004, 0.003, 0.003]
    for s in range(5):
        aa.append([rng[(i + 3 * s) % 20] for i in range(20)])
    make("aa_k2", 20, 2, 1.5, [aa], [7], note="20 states, k=2: 5-bit codes")
