"""`python -m rappas_amd.tools.build`: posterior tables + reference tree -> a phylo-kmer database in the `--jsondb` layout.

The hot loop of the reference's `-p b` phase (src/main_v2/Main_DBBUILD_3.java:648-750) through rk_build_db on the GPU.  What comes
before it in the reference -- running the ancestral-reconstruction tool and parsing its output (src/inputs/*Wrapper.java) -- is not
part of this repository: the input is an .npz with the arrays src/core/PProbasSorted.java holds,
  states u8 [n_nodes, n_sites, n_states], pp_log10 f32 (same shape, descending along the last axis), node_branch u16 [n_nodes],
and optionally gap_off u32 [n_sites + 1] / gap_len i32 (Alignment.getGapIntervals() as CSR; activates gap jumps).
The output is read by `python -m rappas_amd.tools.place --jsondb`.
"""
import argparse
import sys
from types import SimpleNamespace

import numpy as np

from .. import dbbuild, hostio, synth


def build_file(npz, newick, k, omega=1.5, limit_to_1_jump=True, device=0):
    states, pp, nb = npz["states"], npz["pp_log10"], npz["node_branch"]
    alphabet = 4 if states.shape[2] <= 4 else 20
    if alphabet != 4:
        raise ValueError("the --jsondb layout can only carry DNA k-mers (hostio.load_jsondb)")
    thr, thr_log10 = synth.thresholds(omega, alphabet, k)
    gap_off = npz["gap_off"] if "gap_off" in npz else None
    gap_len = npz["gap_len"] if "gap_len" in npz else None
    b = dbbuild.build_db(alphabet, k, states, pp, nb, thr_log10, gap_off=gap_off, gap_len=gap_len,
                         limit_to_1_jump=limit_to_1_jump, device=device)
    tree = hostio.parse_newick(newick)
    if b.branch_ids.size and int(b.branch_ids.max()) >= len(tree.nodes):
        raise ValueError(f"branch id {int(b.branch_ids.max())} does not exist in a tree of {len(tree.nodes)} nodes")
    db = SimpleNamespace(k=k, thr=thr, thr_log10=thr_log10, key_codes=b.key_codes, row_offsets=b.row_offsets,
                         branch_ids=b.branch_ids, scores=b.scores)
    return hostio.dump_jsondb(db, newick, omega=omega), b


def main(argv=None):
    ap = argparse.ArgumentParser(prog="rappas_amd.tools.build", description=__doc__.splitlines()[0])
    ap.add_argument("--pp", required=True, help=".npz with states, pp_log10, node_branch [, gap_off, gap_len]")
    ap.add_argument("--tree", required=True, help="reference tree (Newick); branch ids are its nodes in order of appearance")
    ap.add_argument("-k", type=int, default=8)
    ap.add_argument("--omega", type=float, default=1.5)
    ap.add_argument("--all-jump-combinations", action="store_true", help="gap jumps: not only the first one (limitTo1Jump=false)")
    ap.add_argument("--out", required=True, help="database (JSON, --jsondb layout)")
    ap.add_argument("--device", type=int, default=0)
    a = ap.parse_args(argv)
    with open(a.tree) as f:
        newick = f.read().strip()
    txt, b = build_file(np.load(a.pp), newick, a.k, a.omega, not a.all_jump_combinations, a.device)
    with open(a.out, "w") as f:
        f.write(txt)
    print(f"{b.visits} nodes explored, {b.tuples} words registered -> {len(b.key_codes)} k-mers / {len(b.scores)} entries; "
          f"explore {b.explore_ms:.1f} ms, reduce {b.reduce_ms:.1f} ms -> {a.out}", file=sys.stderr)
    return 0


if __name__ == "__main__":
    sys.exit(main())
