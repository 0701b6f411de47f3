"""ctypes wrapper of the CPU oracle (oracle/rappas_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
PARITY UNPINNED (see rappas_oracle.h): pinned by hand-derived KATs, not by reference outputs.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "liboracle.so")

RO_FLAG_PLACED, RO_FLAG_BAD_CHAR, RO_FLAG_TOO_SHORT, RO_FLAG_AMBIGUOUS, RO_FLAG_BELOW_NSBOUND, RO_FLAG_TIE = 1, 2, 4, 8, 16, 32
AMB_SKIP, AMB_MEAN, AMB_MAX = 0, 1, 2
_lib = None


class ro_counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("reads", "placed", "unplaced", "kmers", "kmers_hit", "entries",
                                         "amb_kmers", "skipped_kmers")]


def build(force=False):
    srcs = [os.path.join(HERE, f) for f in ("rappas_oracle.c", "rappas_oracle.h", "rappas_build_oracle.c", "rappas_build_oracle.h")]
    if force or not os.path.exists(SO) or any(os.path.getmtime(s) > os.path.getmtime(SO) for s in srcs):
        subprocess.run(["make", "-C", HERE, "-B"], check=True, stdout=subprocess.DEVNULL)
    return SO


def load():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(os.environ.get("RO_LIB") or SO)  # RO_LIB: sanitizer build (scripts/asan_host.sh)
        L.ro_thresholds.argtypes = [C.c_float, C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.ro_max_ambig_per_mer.restype = C.c_int
        L.ro_max_ambig_per_mer.argtypes = [C.c_int, C.c_int]
        L.ro_char_code.restype = C.c_uint8
        L.ro_char_code.argtypes = [C.c_int, C.c_int, C.c_uint8]
        L.ro_amb_alternatives.restype = C.c_int
        L.ro_amb_alternatives.argtypes = [C.c_int, C.c_uint8, C.c_void_p]
        L.ro_compress_mer_dna.restype = C.c_int
        L.ro_compress_mer_dna.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.ro_kmer_code.restype = C.c_uint64
        L.ro_kmer_code.argtypes = [C.c_int, C.c_void_p, C.c_int]
        L.ro_db_create.restype = C.c_void_p
        L.ro_db_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_uint64, C.c_void_p,
                                   C.c_void_p, C.c_void_p, C.c_void_p]
        L.ro_db_destroy.argtypes = [C.c_void_p]
        L.ro_place_batch.restype = C.c_int
        L.ro_place_batch.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_float, C.c_uint64, C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.POINTER(ro_counters)]
        L.ro_score_vector.restype = C.c_int
        L.ro_score_vector.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p,
                                      C.POINTER(C.c_int32)]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def thresholds(omega, n_states, k):
    a, b = C.c_float(), C.c_float()
    load().ro_thresholds(omega, n_states, k, C.byref(a), C.byref(b))
    return np.float32(a.value), np.float32(b.value)


def char_code(alphabet, c, convert_uo=False):
    return int(load().ro_char_code(alphabet, int(convert_uo), c if isinstance(c, int) else ord(c)))


def amb_alternatives(alphabet, cls):
    buf = np.zeros(20, np.uint8)
    n = load().ro_amb_alternatives(alphabet, cls, _p(buf))
    return buf[:n].tolist()


def compress_mer_dna(states):
    s = np.ascontiguousarray(states, np.uint8)
    out = np.zeros(16, np.uint8)
    n = load().ro_compress_mer_dna(_p(s), len(s), _p(out))
    return out[:n].tolist()


def kmer_code(alphabet, states):
    s = np.ascontiguousarray(states, np.uint8)
    return int(load().ro_kmer_code(alphabet, _p(s), len(s)))


class OracleDB:
    def __init__(self, alphabet, k, n_branches, thr_log10, thr, key_codes, row_offsets, branch_ids, scores,
                 convert_uo=False):
        self.L = load()
        self.n_branches = n_branches
        kc = np.ascontiguousarray(key_codes, np.uint64)
        ro = np.ascontiguousarray(row_offsets, np.uint64)
        br = np.ascontiguousarray(branch_ids, np.uint16)
        sc = np.ascontiguousarray(scores, np.float32)
        self.h = self.L.ro_db_create(alphabet, int(convert_uo), k, n_branches, float(thr_log10), float(thr),
                                     len(kc), _p(kc), _p(ro), _p(br), _p(sc))
        if not self.h:
            raise ValueError("ro_db_create failed")

    @classmethod
    def from_synth(cls, db, **kw):
        return cls(db.alphabet, db.k, db.n_branches, db.thr_log10, db.thr, db.key_codes, db.row_offsets,
                   db.branch_ids, db.scores, **kw)

    def place(self, seq, seq_off, keep_at_most=7, keep_factor=0.01, amb_mode=AMB_MEAN, ns_bound=float("-inf")):
        seq = np.ascontiguousarray(seq, np.uint8)
        off = np.ascontiguousarray(seq_off, np.uint64)
        n, K = len(off) - 1, keep_at_most
        out = dict(n_rows=np.zeros(n, np.uint8), branch=np.zeros((n, K), np.uint16),
                   score=np.zeros((n, K), np.float32), lwr=np.zeros((n, K), np.float64),
                   flags=np.zeros(n, np.uint32), entries=np.zeros(n, np.uint32))
        ct = ro_counters()
        rc = self.L.ro_place_batch(self.h, K, keep_factor, amb_mode, ns_bound, n, _p(seq), _p(off), _p(out["n_rows"]),
                                   _p(out["branch"]), _p(out["score"]), _p(out["lwr"]), _p(out["flags"]),
                                   _p(out["entries"]), C.byref(ct))
        if rc:
            raise RuntimeError(f"ro_place_batch -> {rc}")
        out["counters"] = {f: getattr(ct, f) for f, _ in ro_counters._fields_}
        return out

    def score_vector(self, read, amb_mode=AMB_MEAN):
        s = np.frombuffer(read if isinstance(read, bytes) else bytes(read), np.uint8)
        S = np.zeros(self.n_branches, np.float32)
        Lo = np.zeros(self.n_branches, np.int32)
        n = C.c_int32()
        f = self.L.ro_score_vector(self.h, amb_mode, _p(np.ascontiguousarray(s)), len(s), _p(S), _p(Lo), C.byref(n))
        return S, Lo[:n.value].copy(), f

    def close(self):
        if self.h:
            self.L.ro_db_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---------------------------------------------------------------------------------------------------------------
# phylo-kmer DB construction (rappas_build_oracle.c): Main_DBBUILD_3.java:648-750 + WordExplorer_v3 + addTuple
# ---------------------------------------------------------------------------------------------------------------
class ro_build_desc(C.Structure):
    _fields_ = [("alphabet", C.c_int), ("k", C.c_int), ("n_nodes", C.c_int), ("n_sites", C.c_int), ("n_states", C.c_int),
                ("states", C.c_void_p), ("pp", C.c_void_p), ("node_branch", C.c_void_p), ("thr_log10", C.c_float),
                ("do_gap_jumps", C.c_int), ("limit_to_1_jump", C.c_int), ("gap_off", C.c_void_p), ("gap_len", C.c_void_p)]


class ro_built(C.Structure):
    _fields_ = [("n_keys", C.c_uint64), ("key_codes", C.POINTER(C.c_uint64)), ("row_offsets", C.POINTER(C.c_uint64)),
                ("branch_ids", C.POINTER(C.c_uint16)), ("scores", C.POINTER(C.c_float)), ("tuples", C.c_uint64),
                ("visits", C.c_uint64)]


def build_db(alphabet, k, states, pp, node_branch, thr_log10, gap_off=None, gap_len=None, limit_to_1_jump=True):
    """states u8 / pp f32: [n_nodes, n_sites, n_states] (rank-ordered, PProbasSorted); node_branch u16 [n_nodes].
    -> dict(key_codes, row_offsets, branch_ids, scores, tuples, visits); rows sorted by code, entries by branch."""
    L = load()
    L.ro_build_db.restype = C.c_int
    L.ro_build_db.argtypes = [C.POINTER(ro_build_desc), C.POINTER(ro_built)]
    L.ro_built_free.argtypes = [C.POINTER(ro_built)]
    states = np.ascontiguousarray(states, np.uint8)
    pp = np.ascontiguousarray(pp, np.float32)
    node_branch = np.ascontiguousarray(node_branch, np.uint16)
    n_nodes, n_sites, n_states = states.shape
    assert pp.shape == states.shape and node_branch.shape == (n_nodes,)
    gaps = gap_off is not None
    if gaps:
        gap_off = np.ascontiguousarray(gap_off, np.uint32)
        gap_len = np.ascontiguousarray(gap_len, np.int32)
        assert gap_off.shape == (n_sites + 1,)
    d = ro_build_desc(alphabet, k, n_nodes, n_sites, n_states, _p(states), _p(pp), _p(node_branch), float(thr_log10),
                      int(gaps), int(bool(limit_to_1_jump)), _p(gap_off) if gaps else None, _p(gap_len) if gaps else None)
    b = ro_built()
    rc = L.ro_build_db(C.byref(d), C.byref(b))
    if rc:
        raise RuntimeError(f"ro_build_db failed: {rc}")
    try:
        nk = int(b.n_keys)
        off = np.ctypeslib.as_array(b.row_offsets, (nk + 1,)).copy()
        ne = int(off[-1])
        return dict(key_codes=np.ctypeslib.as_array(b.key_codes, (max(nk, 1),))[:nk].copy(), row_offsets=off,
                    branch_ids=np.ctypeslib.as_array(b.branch_ids, (max(ne, 1),))[:ne].copy(),
                    scores=np.ctypeslib.as_array(b.scores, (max(ne, 1),))[:ne].copy(), tuples=int(b.tuples), visits=int(b.visits))
    finally:
        L.ro_built_free(C.byref(b))
