"""Host-side mirror of the reference's phylo-kmer construction step, over rk_build_db (include/rappas_place.h).

`build_db` stands where the node / position / first-state loops of src/main_v2/Main_DBBUILD_3.java:648-750 stand (one
src/core/algos/WordExplorer_v3.java explorer per (node, position), registrations through
src/core/hash/CustomHash_v4_FastUtil81.java:73-89); its arguments are what src/core/PProbasSorted.java holds.  No compute happens
in Python and there is no CPU fallback.
"""
import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _lib
from ._lib import rk_build_desc, rk_built_db


@dataclass
class BuiltDB:
    alphabet: int
    k: int
    thr_log10: np.float32
    key_codes: np.ndarray    # u64 [n_keys], ascending
    row_offsets: np.ndarray  # u64 [n_keys + 1]
    branch_ids: np.ndarray   # u16, ascending inside a row
    scores: np.ndarray       # f32
    tuples: int              # registrations before the per-(k-mer, branch) maximum ("Tuples explored")
    visits: int              # explored nodes of the branch-and-bound trees
    explore_ms: float
    reduce_ms: float


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def build_db(alphabet, k, states, pp_log10, node_branch, thr_log10, gap_off=None, gap_len=None, limit_to_1_jump=True, device=0):
    """states u8 / pp_log10 f32: [n_nodes, n_sites, n_states], ranked by descending posterior along the last axis;
    node_branch u16 [n_nodes]; gap_off/gap_len: CSR of Alignment.getGapIntervals() (pass them to activate gap jumps)."""
    lib = _lib.load()
    states = np.ascontiguousarray(states, np.uint8)
    pp = np.ascontiguousarray(pp_log10, np.float32)
    node_branch = np.ascontiguousarray(node_branch, np.uint16)
    if states.ndim != 3 or pp.shape != states.shape or node_branch.shape != (states.shape[0],):
        raise ValueError("states / pp_log10 must be [n_nodes, n_sites, n_states] and node_branch [n_nodes]")
    n_nodes, n_sites, n_states = states.shape
    gaps = gap_off is not None
    if gaps:
        gap_off = np.ascontiguousarray(gap_off, np.uint32)
        gap_len = np.ascontiguousarray(gap_len if gap_len is not None else [], np.int32)
        if gap_off.shape != (n_sites + 1,) or int(gap_off[-1]) != gap_len.shape[0]:
            raise ValueError("gap_off must have n_sites+1 entries and end at len(gap_len)")
    d = rk_build_desc(alphabet, k, n_nodes, n_sites, n_states, int(gaps), int(bool(limit_to_1_jump)), float(thr_log10),
                      _ptr(states), _ptr(pp), _ptr(node_branch), _ptr(gap_off) if gaps else None,
                      _ptr(gap_len) if gaps and gap_len.size else None, device, 0)
    b = rk_built_db()
    _lib.check(lib.rk_build_db(C.byref(d), C.byref(b)))
    try:
        nk, ne = int(b.n_keys), int(b.n_entries)
        take = lambda p, n, dt: (np.ctypeslib.as_array(p, (n,)).copy() if n else np.zeros(0, dt))
        return BuiltDB(alphabet, k, np.float32(thr_log10), take(b.key_codes, nk, np.uint64),
                       np.ctypeslib.as_array(b.row_offsets, (nk + 1,)).copy(), take(b.branch_ids, ne, np.uint16),
                       take(b.scores, ne, np.float32), int(b.tuples), int(b.visits), float(b.explore_ms), float(b.reduce_ms))
    finally:
        lib.rk_built_free(C.byref(b))
