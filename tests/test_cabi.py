"""CPU: the C-ABI shared library loads and exports every symbol include/rappas_place.h declares; argument
validation that needs no GPU.  No compute calls here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import rappas_amd as ra
from rappas_amd import _lib, synth
from oracle import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "rappas_place.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(rk_[a-z_0-9]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    ra.build.build_engine()
    lib = C.CDLL(_lib.lib_path())
    syms = declared_symbols()
    assert len(syms) >= 12
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/rappas_place.h but not exported"
    assert set(syms) == set(_lib.EXPORTS), "ctypes binding and header disagree"


def test_version_and_thresholds():
    lib = _lib.load()
    assert lib.rk_version() == 100
    for ns, k in [(4, 8), (4, 10), (20, 5), (4, 12)]:
        a, b = C.c_float(), C.c_float()
        lib.rk_thresholds(1.5, ns, k, C.byref(a), C.byref(b))
        p, t = O.thresholds(1.5, ns, k)
        assert np.float32(a.value) == p and np.float32(b.value) == t


def test_struct_layouts_match_header():
    # sizes the C compiler gives the header's structs (natural alignment, LP64)
    assert C.sizeof(_lib.rk_db_desc) == 72
    assert C.sizeof(_lib.rk_params) == 16
    assert C.sizeof(_lib.rk_result) == 40
    assert C.sizeof(_lib.rk_counters) == 48
    assert C.sizeof(_lib.rk_db_info) == 80


def test_no_cpu_fallback(gpu_available):
    """Without a device the product path must fail loudly (RK_ERR_NO_DEVICE), never compute on the CPU."""
    if gpu_available:
        pytest.skip("GPU present")
    sdb = synth.make_config_db("C1")
    with pytest.raises(ra.RkError) as ei:
        ra.PhyloKmerDB.from_synth(sdb)
    assert ei.value.code == _lib.RK_ERR_NO_DEVICE
    assert "no CPU fallback" in str(ei.value)


def test_argument_validation_without_device():
    lib = _lib.load()
    h = C.c_void_p()
    assert lib.rk_db_create(None, C.byref(h)) == _lib.RK_ERR_INVALID
    d = _lib.rk_db_desc(alphabet=7, k=8, n_branches=10)
    assert lib.rk_db_create(C.byref(d), C.byref(h)) == _lib.RK_ERR_INVALID
    assert b"alphabet" in lib.rk_last_error()
    d = _lib.rk_db_desc(alphabet=4, k=16, n_branches=10)
    assert lib.rk_db_create(C.byref(d), C.byref(h)) == _lib.RK_ERR_UNSUPPORTED
    d = _lib.rk_db_desc(alphabet=4, k=8, n_branches=70000)
    assert lib.rk_db_create(C.byref(d), C.byref(h)) == _lib.RK_ERR_INVALID
    assert lib.rk_place_batch(None, None, 0, None, None, None, None) == _lib.RK_ERR_INVALID
    assert lib.rk_set_lanes_per_read(None, 16) == _lib.RK_ERR_INVALID
    lib.rk_db_destroy(None)  # no-op


def test_product_package_does_not_import_oracle():
    """The oracle is test infrastructure: nothing under rappas_amd/ or include/ may reference it."""
    for base in ("rappas_amd", "include"):
        for dp, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".hip", ".h", ".cpp", ".c", ".hpp")):
                    txt = open(os.path.join(dp, f)).read()
                    assert "liboracle" not in txt and "rappas_oracle" not in txt and "from oracle" not in txt and "import oracle" not in txt, f
