"""Developer measurement: rk_build_db (GPU) against the CPU oracle on a synthetic posterior table.
usage: build_rate.py [k] [n_nodes] [n_sites] [alphabet] [cpu_nodes]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rappas_amd as ra
from rappas_amd import synth
from oracle import oracle as O

k = int(sys.argv[1]) if len(sys.argv) > 1 else 10
n_nodes = int(sys.argv[2]) if len(sys.argv) > 2 else 64
n_sites = int(sys.argv[3]) if len(sys.argv) > 3 else 600
alphabet = int(sys.argv[4]) if len(sys.argv) > 4 else 4
cpu_nodes = int(sys.argv[5]) if len(sys.argv) > 5 else 1
states, pp, nb = synth.make_pp_tables(alphabet, n_nodes, n_sites, seed=5)
_, T = synth.thresholds(1.5, alphabet, k)
kw = {}
if os.environ.get("GAPS"):  # an alignment with ~30 % gaps: the reference then activates gap jumps (Main_DBBUILD_3.java:239-258)
    rng = np.random.default_rng(9)
    rows = []
    for _ in range(12):
        r = ["A"] * n_sites
        i = 0
        while i < n_sites:
            if rng.random() < 0.06:
                g = int(rng.integers(1, 12))
                for t in range(i, min(n_sites, i + g)):
                    r[t] = "-"
                i += g
            i += 1
        rows.append("".join(r))
    off, lens = synth.gap_intervals(rows)
    kw = dict(gap_off=off, gap_len=lens, limit_to_1_jump=os.environ["GAPS"] != "all")
    print(f"gap jumps on ({'one jump' if kw['limit_to_1_jump'] else 'all combinations'}): {len(lens)} intervals over {n_sites} sites")
ra.build_db(alphabet, 4 if alphabet == 4 else 3, states[:2], pp[:2], nb[:2], T)  # warm-up (module load)
t = time.time()
b = ra.build_db(alphabet, k, states, pp, nb, T, **kw)
wall = time.time() - t
print(f"GPU: alphabet={alphabet} k={k} nodes={n_nodes} sites={n_sites}: visits={b.visits:.4g} tuples={b.tuples:.4g} entries={len(b.scores)} keys={len(b.key_codes)}")
print(f"     explore {b.explore_ms:.1f} ms ({b.visits / b.explore_ms / 1e6:.2f} Gvisits/s), reduce {b.reduce_ms:.1f} ms, wall {wall:.2f} s")
t = time.time()
r = O.build_db(alphabet, k, states[:cpu_nodes], pp[:cpu_nodes], nb[:cpu_nodes], T, **kw)
dt = time.time() - t
print(f"CPU oracle (1 core, {cpu_nodes} node(s)): {r['visits'] / dt / 1e6:.1f} Mvisits/s -> GPU/CPU-core = {b.visits / b.explore_ms * 1e3 / (r['visits'] / dt):.0f}x")
