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
    srcs = [os.path.join(HERE, f) for f in ("rappas_oracle.c", "rappas_oracle.h")]
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
