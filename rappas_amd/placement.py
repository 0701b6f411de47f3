"""Host-side mirror of the reference's placement interface, over the C ABI (include/rappas_place.h).

Names follow the reference: `PhyloKmerDB` stands where `session.hash` (CustomHash_v4_FastUtil81) plus the session
scalars stand (src/main_v2/SessionNext_v2.java:43-66); `PlacementProcess.processQueries` takes the arguments of
src/core/algos/PlacementProcess.java:471-483 that reach the hot path (keepAtMost, keepFactor, treatAmbiguities,
treatAmbiguitiesWithMax) and returns, per read, what :974-1025 turns into jplace rows.
No compute happens in Python and nothing here falls back to a CPU implementation.
"""
import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _lib
from ._lib import (RK_ALPHABET_AA, RK_ALPHABET_DNA, RK_AMB_MAX, RK_AMB_MEAN, RK_AMB_SKIP, RK_TABLE_AUTO,
                   RK_TABLE_DIRECT, RK_TABLE_DIRECT8, RK_TABLE_HASH, rk_counters, rk_db_desc, rk_db_info, rk_params, rk_result)


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


@dataclass
class Placements:
    """Rows best -> worse per read; unused rows are branch 0xFFFF / score -inf / lwr 0."""
    n_rows: np.ndarray   # u8  [n]
    branch: np.ndarray   # u16 [n, K]
    score: np.ndarray    # f32 [n, K]
    lwr: np.ndarray      # f64 [n, K]
    flags: np.ndarray    # u32 [n]
    counters: dict


def pack_reads(alphabet, k, seq, seq_off, words_per_read=None, convert_uo=False, threads=0):
    """rk_pack_reads: the host-side read packer without a database handle (no GPU): ASCII reads -> (packed u32 [n, wpr], lens u32 [n],
    flags u32 [n]), the records the device packer produces (AmbigSequenceKnife.java:103-130 char -> state)."""
    lib = _lib.load()
    seq = np.ascontiguousarray(seq, dtype=np.uint8)
    seq_off = np.ascontiguousarray(seq_off, dtype=np.uint64)
    n = seq_off.shape[0] - 1
    bits = 2 if alphabet == _lib.RK_ALPHABET_DNA else 5
    if words_per_read is None:
        max_len = int((seq_off[1:] - seq_off[:-1]).max()) if n else 0
        words_per_read = max(1, (max_len * bits + 31) // 32)
    packed = np.zeros((n, words_per_read), np.uint32)
    lens = np.zeros(n, np.uint32)
    flags = np.zeros(n, np.uint32)
    _lib.check(lib.rk_pack_reads(alphabet, int(bool(convert_uo)), k, n, _ptr(seq), _ptr(seq_off), words_per_read, _ptr(packed), _ptr(lens), _ptr(flags), threads))
    return packed, lens, flags


def host_alloc(shape, dtype):
    """numpy array in page-locked host memory (rk_host_alloc): buffers the DMA reads / writes directly, no staging copies in
    rk_place_batch / rk_place_batch_packed.  The memory lives until the process ends (tests and the bench allocate a handful)."""
    lib = _lib.load()
    shape = (shape,) if np.isscalar(shape) else tuple(shape)
    nbytes = max(1, int(np.prod(shape)) * np.dtype(dtype).itemsize)
    p = lib.rk_host_alloc(nbytes)
    if not p:
        raise _lib.RkError(_lib.RK_ERR_NOMEM, lib.rk_last_error().decode("utf-8", "replace"))
    return np.ctypeslib.as_array((C.c_uint8 * nbytes).from_address(p)).view(dtype)[:int(np.prod(shape))].reshape(shape)


def validate_db(alphabet, k, n_branches, thr_log10, thr, key_codes, row_offsets, branch_ids, scores,
                table_mode=RK_TABLE_AUTO, convert_uo=False):
    """rk_db_validate: argument checks + host-side image construction, no device needed. Returns rk_db_info."""
    lib = _lib.load()
    key_codes = np.ascontiguousarray(key_codes, dtype=np.uint64)
    row_offsets = np.ascontiguousarray(row_offsets, dtype=np.uint64)
    branch_ids = np.ascontiguousarray(branch_ids, dtype=np.uint16)
    scores = np.ascontiguousarray(scores, dtype=np.float32)
    d = rk_db_desc(alphabet, int(bool(convert_uo)), k, n_branches, float(thr_log10), float(thr), key_codes.shape[0],
                   _ptr(key_codes), _ptr(row_offsets), _ptr(branch_ids), _ptr(scores), 0, table_mode)
    info = rk_db_info()
    _lib.check(lib.rk_db_validate(C.byref(d), C.byref(info)))
    return info


def save_db_image(path, alphabet, k, n_branches, thr_log10, thr, key_codes, row_offsets, branch_ids, scores, table_mode=RK_TABLE_AUTO,
                  convert_uo=False, user=b""):
    """rk_db_save_desc: the image file of a database given as CSR arrays, built on the host (no GPU needed)."""
    lib = _lib.load()
    key_codes = np.ascontiguousarray(key_codes, dtype=np.uint64)
    row_offsets = np.ascontiguousarray(row_offsets, dtype=np.uint64)
    branch_ids = np.ascontiguousarray(branch_ids, dtype=np.uint16)
    scores = np.ascontiguousarray(scores, dtype=np.float32)
    d = rk_db_desc(alphabet, int(bool(convert_uo)), k, n_branches, float(thr_log10), float(thr), key_codes.shape[0],
                   _ptr(key_codes), _ptr(row_offsets), _ptr(branch_ids), _ptr(scores), 0, table_mode)
    user = bytes(user)
    _lib.check(lib.rk_db_save_desc(C.byref(d), str(path).encode(), user, len(user)))


def db_image_info(path):
    """rk_db_image_info + rk_db_image_user: (rk_db_info, user blob) of an image file after its size / checksum tests; no GPU needed."""
    lib = _lib.load()
    info, n = rk_db_info(), C.c_uint64(0)
    _lib.check(lib.rk_db_image_info(str(path).encode(), C.byref(info), C.byref(n)))
    buf = C.create_string_buffer(max(1, n.value))
    _lib.check(lib.rk_db_image_user(str(path).encode(), buf, n.value, C.byref(n)))
    return info, buf.raw[:n.value]


class PhyloKmerDB:
    """Phylo-kmer DB resident in one GPU's HBM (open-addressed / direct table + CSR rows)."""

    def __init__(self, alphabet, k, n_branches, thr_log10, thr, key_codes, row_offsets, branch_ids, scores,
                 device=0, table_mode=RK_TABLE_AUTO, convert_uo=False):
        self._lib = _lib.load()
        self._h = C.c_void_p()
        key_codes = np.ascontiguousarray(key_codes, dtype=np.uint64)
        row_offsets = np.ascontiguousarray(row_offsets, dtype=np.uint64)
        branch_ids = np.ascontiguousarray(branch_ids, dtype=np.uint16)
        scores = np.ascontiguousarray(scores, dtype=np.float32)
        if row_offsets.shape[0] != key_codes.shape[0] + 1:
            raise ValueError("row_offsets must have n_keys+1 entries")
        if key_codes.shape[0] and int(row_offsets[-1]) != branch_ids.shape[0]:
            raise ValueError("row_offsets[-1] must equal len(branch_ids)")
        if branch_ids.shape[0] != scores.shape[0]:
            raise ValueError("branch_ids and scores differ in length")
        d = rk_db_desc(alphabet, int(bool(convert_uo)), k, n_branches, float(thr_log10), float(thr),
                       key_codes.shape[0], _ptr(key_codes), _ptr(row_offsets), _ptr(branch_ids), _ptr(scores),
                       device, table_mode)
        _lib.check(self._lib.rk_db_create(C.byref(d), C.byref(self._h)))
        info = rk_db_info()
        _lib.check(self._lib.rk_db_get_info(self._h, C.byref(info)))
        self.info = info

    @classmethod
    def synthetic(cls, spec, device=0, table_mode=RK_TABLE_AUTO, convert_uo=False):
        """The seeded synthetic database of SURVEY 8(d) generated on the device straight into the HBM image
        (rk_db_create_synth); `spec` is a rappas_amd.synth.SynthSpec, whose numpy twin regenerates any row on the host."""
        self = cls.__new__(cls)
        self._lib = _lib.load()
        self._h = C.c_void_p()
        d = _lib.rk_synth_desc(spec.alphabet, int(bool(convert_uo)), spec.k, spec.n_branches, float(spec.thr_log10), float(spec.thr),
                               int(spec.seed) & 0xFFFFFFFFFFFFFFFF, float(spec.key_fraction), float(spec.mean_row_len), device, table_mode)
        _lib.check(self._lib.rk_db_create_synth(C.byref(d), C.byref(self._h)))
        info = rk_db_info()
        _lib.check(self._lib.rk_db_get_info(self._h, C.byref(info)))
        self.info = info
        return self

    def save(self, path, user=b""):
        """rk_db_save: this handle's HBM image as a file (the reference's SessionNext_v2.storeHash, SessionNext_v2.java:110-154);
        `user` = bytes the caller wants next to it (the tools keep the reference tree there)."""
        user = bytes(user)
        _lib.check(self._lib.rk_db_save(self.handle, str(path).encode(), user, len(user)))

    @classmethod
    def load(cls, path, device=0):
        """rk_db_load: a handle from an image file -- mmap + one upload per section, no rebuild (SessionNext_v2.load, :158-207)."""
        self = cls.__new__(cls)
        self._lib = _lib.load()
        self._h = C.c_void_p()
        _lib.check(self._lib.rk_db_load(str(path).encode(), device, C.byref(self._h)))
        info = rk_db_info()
        _lib.check(self._lib.rk_db_get_info(self._h, C.byref(info)))
        self.info = info
        return self

    def clone(self, device=0):
        """rk_db_clone: another handle of this database on `device`, copied device to device (no rebuild, no host round trip)."""
        other = type(self).__new__(type(self))
        other._lib = self._lib
        other._h = C.c_void_p()
        _lib.check(self._lib.rk_db_clone(self.handle, device, C.byref(other._h)))
        info = rk_db_info()
        _lib.check(self._lib.rk_db_get_info(other._h, C.byref(info)))
        other.info = info
        return other

    def fetch_row(self, code):
        """(branch_ids u16[len], scores f32[len]) of one k-mer code as stored in the HBM image; empty arrays if absent
        (CustomHash_v4_FastUtil81.getPairsOfTopPosition2, src/core/hash/CustomHash_v4_FastUtil81.java:146-153)."""
        n = C.c_uint32(0)
        cap = int(self.info.max_row_len)
        br = np.zeros(max(cap, 1), np.uint16)
        sc = np.zeros(max(cap, 1), np.float32)
        _lib.check(self._lib.rk_db_fetch_row(self.handle, int(code), cap, C.byref(n), _ptr(br), _ptr(sc)))
        return br[:n.value].copy(), sc[:n.value].copy()

    @classmethod
    def from_synth(cls, db, **kw):
        return cls(db.alphabet, db.k, db.n_branches, db.thr_log10, db.thr, db.key_codes, db.row_offsets,
                   db.branch_ids, db.scores, **kw)

    @property
    def handle(self):
        if not self._h:
            raise RuntimeError("PhyloKmerDB is closed")
        return self._h

    def set_lanes_per_read(self, lanes):
        _lib.check(self._lib.rk_set_lanes_per_read(self.handle, lanes))

    def kernel_name(self):
        return self._lib.rk_kernel_name(self.handle).decode()

    def packed_words(self, max_len):
        return int(self._lib.rk_packed_words(self.handle, max_len))

    def close(self):
        if getattr(self, "_h", None):
            self._lib.rk_db_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class PlacementProcess:
    """Mirror of core.algos.PlacementProcess for the hot path."""

    def __init__(self, db, ns_bound=float("-inf")):
        self.db = db
        self.ns_bound = ns_bound  # PlacementProcess(session, nsBound, queryLimit) (Main_PLACEMENT_v07.java:248-253)
        self._lib = _lib.load()

    def _params(self, keepAtMost, keepFactor, treatAmbiguities, treatAmbiguitiesWithMax):
        amb = RK_AMB_SKIP if not treatAmbiguities else (RK_AMB_MAX if treatAmbiguitiesWithMax else RK_AMB_MEAN)
        return rk_params(keepAtMost, keepFactor, amb, self.ns_bound)

    def processQueries(self, seq, seq_off, keepAtMost=7, keepFactor=0.01, treatAmbiguities=True,
                       treatAmbiguitiesWithMax=False, out=None):
        """seq: uint8 ASCII of all reads concatenated (no gap stripping, as FASTAPointer(q,false) delivers them);
        seq_off: uint64 [n+1].  Defaults = src/main_v2/ArgumentsParser_v2.java:87-91.  `out`: a Placements to reuse."""
        seq = np.ascontiguousarray(seq, dtype=np.uint8)
        seq_off = np.ascontiguousarray(seq_off, dtype=np.uint64)
        n = seq_off.shape[0] - 1
        K = keepAtMost
        if out is None:
            out = Placements(np.zeros(n, np.uint8), np.zeros((n, K), np.uint16), np.zeros((n, K), np.float32),
                             np.zeros((n, K), np.float64), np.zeros(n, np.uint32), {})
        res = rk_result(_ptr(out.n_rows), _ptr(out.branch), _ptr(out.score), _ptr(out.lwr), _ptr(out.flags))
        p = self._params(keepAtMost, keepFactor, treatAmbiguities, treatAmbiguitiesWithMax)
        ct = rk_counters()
        _lib.check(self._lib.rk_place_batch(self.db.handle, C.byref(p), n, _ptr(seq), _ptr(seq_off), C.byref(res),
                                            C.byref(ct)))
        out.counters = {f: getattr(ct, f) for f, _ in rk_counters._fields_}
        return out

    def pack_reads_host(self, seq, seq_off, max_len=None, threads=0, out=None):
        """rk_pack_reads_host: ASCII reads -> (packed u32 [n, wpr], lens u32 [n], flags u32 [n]) on the host, the records the
        device packer would produce (AmbigSequenceKnife.java:103-130 char -> state).  `out` = (packed, lens, flags) to reuse."""
        seq = np.ascontiguousarray(seq, dtype=np.uint8)
        seq_off = np.ascontiguousarray(seq_off, dtype=np.uint64)
        n = seq_off.shape[0] - 1
        if max_len is None:
            max_len = int((seq_off[1:] - seq_off[:-1]).max()) if n else 0
        wpr = self.db.packed_words(max_len)
        if out is not None:
            packed, lens, flags = out
            assert packed.shape == (n, wpr) and packed.dtype == np.uint32 and lens.shape == (n,) and flags.shape == (n,)
        else:
            packed = np.zeros((n, wpr), np.uint32)
            lens = np.zeros(n, np.uint32)
            flags = np.zeros(n, np.uint32)
        _lib.check(self._lib.rk_pack_reads_host(self.db.handle, n, _ptr(seq), _ptr(seq_off), wpr, _ptr(packed), _ptr(lens), _ptr(flags), threads))
        return packed, lens, flags

    def processQueriesPacked(self, packed, lens=None, fixed_len=0, flags=None, seq=None, seq_off=None, keepAtMost=7, keepFactor=0.01,
                             treatAmbiguities=True, treatAmbiguitiesWithMax=False, out=None):
        """rk_place_batch_packed: processQueries for reads already packed on the host (38 instead of 150 bytes per 150-bp read over
        PCIe); seq / seq_off are only needed for reads flagged AMBIGUOUS.  `out`: a Placements whose arrays are reused."""
        packed = np.ascontiguousarray(packed, dtype=np.uint32)
        n, wpr = packed.shape
        K = keepAtMost
        if out is None:
            out = Placements(np.zeros(n, np.uint8), np.zeros((n, K), np.uint16), np.zeros((n, K), np.float32),
                             np.zeros((n, K), np.float64), np.zeros(n, np.uint32), {})
        res = rk_result(_ptr(out.n_rows), _ptr(out.branch), _ptr(out.score), _ptr(out.lwr), _ptr(out.flags))
        p = self._params(keepAtMost, keepFactor, treatAmbiguities, treatAmbiguitiesWithMax)
        ct = rk_counters()
        keep = [np.ascontiguousarray(a, dtype=dt) if a is not None else None
                for a, dt in ((lens, np.uint32), (flags, np.uint32), (seq, np.uint8), (seq_off, np.uint64))]
        ptrs = [None if a is None else _ptr(a) for a in keep]
        _lib.check(self._lib.rk_place_batch_packed(self.db.handle, C.byref(p), n, _ptr(packed), wpr, ptrs[0], fixed_len, ptrs[1],
                                                   ptrs[2], ptrs[3], C.byref(res), C.byref(ct)))
        out.counters = {f: getattr(ct, f) for f, _ in rk_counters._fields_}
        return out

    def processQueriesMulti(self, dbs, seq, seq_off, keepAtMost=7, keepFactor=0.01, treatAmbiguities=True,
                            treatAmbiguitiesWithMax=False, out=None):
        """processQueries over several device handles of the same database from this one process
        (rk_place_batch_multi: contiguous shards, one host thread per handle, no collective)."""
        seq = np.ascontiguousarray(seq, dtype=np.uint8)
        seq_off = np.ascontiguousarray(seq_off, dtype=np.uint64)
        n = seq_off.shape[0] - 1
        K = keepAtMost
        if out is None:
            out = Placements(np.zeros(n, np.uint8), np.zeros((n, K), np.uint16), np.zeros((n, K), np.float32),
                             np.zeros((n, K), np.float64), np.zeros(n, np.uint32), {})
        res = rk_result(_ptr(out.n_rows), _ptr(out.branch), _ptr(out.score), _ptr(out.lwr), _ptr(out.flags))
        p = self._params(keepAtMost, keepFactor, treatAmbiguities, treatAmbiguitiesWithMax)
        ct = rk_counters()
        handles = (C.c_void_p * len(dbs))(*[d.handle for d in dbs])
        _lib.check(self._lib.rk_place_batch_multi(handles, len(dbs), C.byref(p), n, _ptr(seq), _ptr(seq_off), C.byref(res),
                                                  C.byref(ct)))
        out.counters = {f: getattr(ct, f) for f, _ in rk_counters._fields_}
        return out

    # ---- device-resident variant (torch tensors only carry the memory and the stream) ----
    def place_packed(self, packed, fixed_len=0, lens=None, flags_in=None, seq_ascii=None, seq_off=None, out=None,
                     keepAtMost=7, keepFactor=0.01, treatAmbiguities=True, treatAmbiguitiesWithMax=False,
                     stream=None):
        import torch
        n, wpr = packed.shape
        dev = packed.device
        K = keepAtMost
        if out is None:
            out = dict(n_rows=torch.empty(n, dtype=torch.uint8, device=dev),
                       branch=torch.empty((n, K), dtype=torch.int16, device=dev),
                       score=torch.empty((n, K), dtype=torch.float32, device=dev),
                       lwr=torch.empty((n, K), dtype=torch.float64, device=dev),
                       flags=torch.empty(n, dtype=torch.int32, device=dev))
        res = rk_result(out["n_rows"].data_ptr(), out["branch"].data_ptr(), out["score"].data_ptr(),
                        out["lwr"].data_ptr(), out["flags"].data_ptr())
        p = self._params(keepAtMost, keepFactor, treatAmbiguities, treatAmbiguitiesWithMax)
        st = stream if stream is not None else torch.cuda.current_stream(dev).cuda_stream
        dp = lambda t: None if t is None else t.data_ptr()
        _lib.check(self._lib.rk_place_packed_device(self.db.handle, C.byref(p), n, packed.data_ptr(), wpr, dp(lens),
                                                    fixed_len, dp(flags_in), dp(seq_ascii), dp(seq_off),
                                                    C.byref(res), C.c_void_p(st)))
        return out

    def count_work(self, packed, fixed_len=0, lens=None, flags_in=None, stream=None):
        """k-mers probed / k-mers with a row / row entries walked for a batch of packed reads (rk_count_work_device: a kernel of its own,
        the placement kernels carry no counters) -> dict"""
        import torch
        n, wpr = packed.shape
        dev = packed.device
        out = torch.empty(3, dtype=torch.int64, device=dev)
        st = stream if stream is not None else torch.cuda.current_stream(dev).cuda_stream
        dp = lambda t: None if t is None else t.data_ptr()
        _lib.check(self._lib.rk_count_work_device(self.db.handle, n, packed.data_ptr(), wpr, dp(lens), fixed_len, dp(flags_in), out.data_ptr(), C.c_void_p(st)))
        if stream is not None:
            torch.cuda.synchronize(dev)
        probed, hit, entries = (int(x) for x in out.tolist())
        return {"kmers_probed": probed, "kmers_hit": hit, "entries": entries}

    def pack_reads(self, seq_ascii, seq_off, max_len, stream=None):
        import torch
        n = seq_off.shape[0] - 1
        dev = seq_ascii.device
        wpr = self.db.packed_words(max_len)
        packed = torch.empty((n, wpr), dtype=torch.int32, device=dev)
        lens = torch.empty(n, dtype=torch.int32, device=dev)
        flags = torch.empty(n, dtype=torch.int32, device=dev)
        st = stream if stream is not None else torch.cuda.current_stream(dev).cuda_stream
        _lib.check(self._lib.rk_pack_reads_device(self.db.handle, n, seq_ascii.data_ptr(), seq_off.data_ptr(), wpr,
                                                  packed.data_ptr(), lens.data_ptr(), flags.data_ptr(),
                                                  C.c_void_p(st)))
        return packed, lens, flags
