"""ctypes binding of the C ABI in include/rappas_place.h (the product boundary)."""
import ctypes as C
import os

from . import build as _build

_LIB = None

RK_ALPHABET_DNA, RK_ALPHABET_AA = 4, 20
RK_AMB_SKIP, RK_AMB_MEAN, RK_AMB_MAX = 0, 1, 2
RK_TABLE_AUTO, RK_TABLE_HASH, RK_TABLE_DIRECT, RK_TABLE_DIRECT8 = 0, 1, 2, 4
RK_FLAG_PLACED, RK_FLAG_BAD_CHAR, RK_FLAG_TOO_SHORT, RK_FLAG_AMBIGUOUS, RK_FLAG_BELOW_NSBOUND, RK_FLAG_TOO_LONG = 1, 2, 4, 8, 16, 64
RK_OK, RK_ERR_INVALID, RK_ERR_NO_DEVICE, RK_ERR_HIP, RK_ERR_NOMEM, RK_ERR_UNSUPPORTED, RK_ERR_IO = 0, -1, -2, -3, -4, -5, -6


class rk_db_desc(C.Structure):
    _fields_ = [
        ("alphabet", C.c_uint32), ("convert_uo", C.c_uint32), ("k", C.c_uint32), ("n_branches", C.c_uint32),
        ("thr_log10", C.c_float), ("thr", C.c_float), ("n_keys", C.c_uint64),
        ("key_codes", C.c_void_p), ("row_offsets", C.c_void_p), ("branch_ids", C.c_void_p), ("scores", C.c_void_p),
        ("device", C.c_int32), ("table_mode", C.c_uint32),
    ]


class rk_db_info(C.Structure):
    _fields_ = [
        ("alphabet", C.c_uint32), ("k", C.c_uint32), ("n_branches", C.c_uint32), ("table_mode", C.c_uint32),
        ("thr_log10", C.c_float), ("thr", C.c_float), ("n_keys", C.c_uint64), ("n_entries", C.c_uint64),
        ("table_slots", C.c_uint64), ("table_bytes", C.c_uint64), ("rows_bytes", C.c_uint64),
        ("bits_per_symbol", C.c_uint32), ("max_row_len", C.c_uint32), ("device", C.c_int32),
    ]


class rk_params(C.Structure):
    _fields_ = [("keep_at_most", C.c_uint32), ("keep_factor", C.c_float), ("amb_mode", C.c_uint32), ("ns_bound", C.c_float)]


class rk_counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("reads", "placed", "unplaced", "bad_char", "too_short", "ambiguous")]


class rk_result(C.Structure):
    _fields_ = [("n_rows", C.c_void_p), ("branch", C.c_void_p), ("score", C.c_void_p), ("lwr", C.c_void_p), ("flags", C.c_void_p)]


class rk_synth_desc(C.Structure):
    _fields_ = [
        ("alphabet", C.c_uint32), ("convert_uo", C.c_uint32), ("k", C.c_uint32), ("n_branches", C.c_uint32),
        ("thr_log10", C.c_float), ("thr", C.c_float), ("seed", C.c_uint64), ("key_fraction", C.c_double),
        ("mean_row_len", C.c_double), ("device", C.c_int32), ("table_mode", C.c_uint32),
    ]


class rk_build_desc(C.Structure):
    _fields_ = [
        ("alphabet", C.c_uint32), ("k", C.c_uint32), ("n_nodes", C.c_uint32), ("n_sites", C.c_uint32), ("n_states", C.c_uint32),
        ("do_gap_jumps", C.c_uint32), ("limit_to_1_jump", C.c_uint32), ("thr_log10", C.c_float),
        ("states", C.c_void_p), ("pp_log10", C.c_void_p), ("node_branch", C.c_void_p), ("gap_off", C.c_void_p),
        ("gap_len", C.c_void_p), ("device", C.c_int32), ("reserved", C.c_uint32),
    ]


class rk_built_db(C.Structure):
    _fields_ = [
        ("n_keys", C.c_uint64), ("n_entries", C.c_uint64), ("key_codes", C.POINTER(C.c_uint64)),
        ("row_offsets", C.POINTER(C.c_uint64)), ("branch_ids", C.POINTER(C.c_uint16)), ("scores", C.POINTER(C.c_float)),
        ("tuples", C.c_uint64), ("visits", C.c_uint64), ("explore_ms", C.c_double), ("reduce_ms", C.c_double),
    ]


# every symbol include/rappas_place.h declares
EXPORTS = {
    "rk_version": (C.c_int, []),
    "rk_last_error": (C.c_char_p, []),
    "rk_thresholds": (None, [C.c_float, C.c_uint32, C.c_uint32, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "rk_db_create": (C.c_int, [C.POINTER(rk_db_desc), C.POINTER(C.c_void_p)]),
    "rk_db_validate": (C.c_int, [C.POINTER(rk_db_desc), C.POINTER(rk_db_info)]),
    "rk_db_destroy": (None, [C.c_void_p]),
    "rk_db_clone": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_void_p)]),
    "rk_db_get_info": (C.c_int, [C.c_void_p, C.POINTER(rk_db_info)]),
    "rk_db_save": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_uint64]),
    "rk_db_save_desc": (C.c_int, [C.POINTER(rk_db_desc), C.c_char_p, C.c_void_p, C.c_uint64]),
    "rk_db_load": (C.c_int, [C.c_char_p, C.c_int32, C.POINTER(C.c_void_p)]),
    "rk_db_image_info": (C.c_int, [C.c_char_p, C.POINTER(rk_db_info), C.POINTER(C.c_uint64)]),
    "rk_db_image_user": (C.c_int, [C.c_char_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]),
    "rk_db_fetch_row": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint32, C.POINTER(C.c_uint32), C.c_void_p, C.c_void_p]),
    "rk_db_create_synth": (C.c_int, [C.POINTER(rk_synth_desc), C.POINTER(C.c_void_p)]),
    "rk_place_batch": (C.c_int, [C.c_void_p, C.POINTER(rk_params), C.c_uint64, C.c_void_p, C.c_void_p,
                                 C.POINTER(rk_result), C.POINTER(rk_counters)]),
    "rk_reserve_host_path": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32]),
    "rk_place_batch_packed": (C.c_int, [C.c_void_p, C.POINTER(rk_params), C.c_uint64, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(rk_result), C.POINTER(rk_counters)]),
    "rk_pack_reads_host": (C.c_int, [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_uint32]),
    "rk_pack_reads": (C.c_int, [C.c_uint32, C.c_int, C.c_uint32, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p,
                                C.c_uint32]),
    "rk_place_batch_multi": (C.c_int, [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(rk_params), C.c_uint64, C.c_void_p,
                                       C.c_void_p, C.POINTER(rk_result), C.POINTER(rk_counters)]),
    "rk_host_alloc": (C.c_void_p, [C.c_uint64]),
    "rk_host_free": (None, [C.c_void_p]),
    "rk_packed_words": (C.c_uint32, [C.c_void_p, C.c_uint32]),
    "rk_pack_reads_device": (C.c_int, [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_void_p]),
    "rk_place_packed_device": (C.c_int, [C.c_void_p, C.POINTER(rk_params), C.c_uint64, C.c_void_p, C.c_uint32,
                                         C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.POINTER(rk_result), C.c_void_p]),
    "rk_count_work_device": (C.c_int, [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "rk_set_lanes_per_read": (C.c_int, [C.c_void_p, C.c_uint32]),
    "rk_kernel_name": (C.c_char_p, [C.c_void_p]),
    "rk_build_db": (C.c_int, [C.POINTER(rk_build_desc), C.POINTER(rk_built_db)]),
    "rk_built_free": (None, [C.POINTER(rk_built_db)]),
}


def lib_path():
    # RK_LIB: developer override (timing-only ablation builds); the default is the in-tree product library
    return os.environ.get("RK_LIB") or _build.ENGINE_SO


_DEV = None


def load_dev():
    """The -DRK_DEV_KNOBS build (librappas_place_dev.so): the only build that reads the developer / test environment knobs.
    Tests that need a knob switch to it with the `dev_lib` fixture (tests/conftest.py); nothing in the package does."""
    global _DEV
    if _DEV is None:
        _DEV = _open(os.environ.get("RK_LIB") or _build.DEV_SO, C.RTLD_LOCAL)
    return _DEV


def load():
    """Load librappas_place.so.  Fails loudly if the HIP extension has not been built: no fallback."""
    global _LIB
    if _LIB is not None:
        return _LIB
    _LIB = _open(lib_path(), C.RTLD_GLOBAL)
    return _LIB


def _open(path, mode):
    if not os.path.exists(path):
        raise RuntimeError(f"{path} is missing: build the gfx950 engine first (python -m rappas_amd.build); "
                           "there is no CPU fallback for the placement path")
    try:
        # torch bundles its own libamdhip64.so.7; importing it first makes the process share ONE HIP runtime
        # with the tensors whose device pointers we hand to the engine.
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - torch is plumbing, the engine itself does not need it
        pass
    lib = C.CDLL(path, mode=mode)
    for name, (res, args) in EXPORTS.items():
        if os.environ.get("RK_LIB") and not hasattr(lib, name):
            continue  # developer timing builds (scripts/) compile the placement unit only, or an earlier revision of it
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return lib


class RkError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"rk error {code}: {msg}")
        self.code = code


def check(rc):
    if rc != RK_OK:
        raise RkError(rc, load().rk_last_error().decode("utf-8", "replace"))
