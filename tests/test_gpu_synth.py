"""GPU (-m gpu): the device-side synthetic database (rk_db_create_synth) -- what BASELINE config C5 (~200 GB) is built with.

The image is generated in HBM from (seed, k-mer index, entry index); rappas_amd.synth.SynthSpec regenerates the same rows
with numpy.  Checks: rows read back through rk_db_fetch_row equal the numpy twin's; placements against the generated image
equal the oracle's on a database rebuilt host-side from the rows the reads touch; and the same at C5's full size (an
image of ~200 GB, far beyond what a host array could carry)."""
import numpy as np
import pytest

import rappas_amd as ra
from rappas_amd import synth
from oracle import oracle as O
from tests.util import compare_with_oracle

pytestmark = pytest.mark.gpu

MODES = {"direct": ra.RK_TABLE_DIRECT, "direct8": ra.RK_TABLE_DIRECT8, "hash": ra.RK_TABLE_HASH}


def small_spec(alphabet=4, k=8, n_branches=999, mean=13.0, kf=0.75, seed=7):
    thr, t = synth.thresholds(1.5, alphabet, k)
    return synth.SynthSpec(alphabet, k, n_branches, thr, t, seed, kf, mean)


def check_rows(db, spec, dense):
    present, lens = spec.present_and_lens(dense)
    codes = synth.dense_to_code(spec.alphabet, spec.k, dense)
    for i, d in enumerate(dense):
        br, sc = db.fetch_row(int(codes[i]))
        if not present[i]:
            assert len(br) == 0
            continue
        off, wb, ws = spec.rows(np.array([d], np.uint64))
        assert len(br) == int(lens[i]) == len(wb)
        assert np.array_equal(br, wb)
        assert np.array_equal(sc.view(np.uint32), ws.view(np.uint32))


@pytest.mark.parametrize("table", ["direct", "direct8", "hash"])
@pytest.mark.parametrize("shape", ["small_tree", "large_tree", "protein"])
def test_device_rows_equal_the_host_twin(table, shape):
    spec = {"small_tree": small_spec(), "large_tree": small_spec(n_branches=19_999, mean=300.0, k=7),
            "protein": small_spec(alphabet=20, k=3, n_branches=399, mean=9.0, kf=0.4)}[shape]
    db = ra.PhyloKmerDB.synthetic(spec, table_mode=MODES[table])
    try:
        rng = np.random.default_rng(3)
        dense = np.unique(np.concatenate([rng.integers(0, spec.space, 60), [0, spec.space - 1]])).astype(np.uint64)
        check_rows(db, spec, dense)
        tab = spec.row_length_table()
        assert db.info.n_keys == int((tab > 0).sum()) and db.info.n_entries == int(tab.sum())
        assert db.info.max_row_len == int(tab.max())
    finally:
        db.close()


@pytest.mark.parametrize("shape", ["small_tree", "large_tree", "protein"])
@pytest.mark.parametrize("amb", ["mean", "max"])
def test_placements_against_generated_image_equal_oracle(shape, amb):
    """whole k-mer space rebuilt on the host (small k), so reads with ambiguity characters are covered too"""
    spec = {"small_tree": small_spec(), "large_tree": small_spec(n_branches=19_999, mean=300.0, k=7),
            "protein": small_spec(alphabet=20, k=3, n_branches=399, mean=9.0, kf=0.4)}[shape]
    sdb = spec.subset_db(np.arange(spec.space, dtype=np.uint64))
    odb = O.OracleDB.from_synth(sdb)
    rlen = 100 if spec.alphabet == 20 else 150
    seq, off = synth.make_reads(spec.alphabet, 600, rlen, seed=5, amb_rate=0.003, var_len=30)
    db = ra.PhyloKmerDB.synthetic(spec)
    try:
        got = ra.PlacementProcess(db).processQueries(seq, off, treatAmbiguitiesWithMax=amb == "max")
        mode = O.AMB_MAX if amb == "max" else O.AMB_MEAN
        ref = odb.place(seq, off, amb_mode=mode)
        st = compare_with_oracle(got, ref, odb, seq, off, amb_mode=mode)
        assert st["placed"] > 500
    finally:
        db.close()


def test_generated_image_equals_rk_db_create_of_the_same_rows():
    """the same database through the host-array entry point gives identical placements (two image builders, one result)"""
    spec = small_spec(k=9, n_branches=2_001, mean=20.0)
    sdb = spec.subset_db(np.arange(spec.space, dtype=np.uint64))
    seq, off = synth.make_reads(4, 3000, 150, seed=2)
    a = ra.PhyloKmerDB.synthetic(spec)
    b = ra.PhyloKmerDB.from_synth(sdb)
    try:
        ga = ra.PlacementProcess(a).processQueries(seq, off)
        gb = ra.PlacementProcess(b).processQueries(seq, off)
        assert a.info.n_entries == b.info.n_entries and a.info.rows_bytes == b.info.rows_bytes
        for f in ("n_rows", "branch", "flags"):
            assert np.array_equal(getattr(ga, f), getattr(gb, f))
        assert np.array_equal(ga.score.view(np.uint32), gb.score.view(np.uint32)) and np.array_equal(ga.lwr, gb.lwr)
    finally:
        a.close()
        b.close()


def test_argument_errors():
    import ctypes as C
    from rappas_amd import _lib
    lib = _lib.load()
    h = C.c_void_p()
    bad = _lib.rk_synth_desc(4, 0, 8, 999, -3.0, 0.001, 1, 0.0, 10.0, 0, 0)
    assert lib.rk_db_create_synth(C.byref(bad), C.byref(h)) == _lib.RK_ERR_INVALID
    bad = _lib.rk_synth_desc(4, 0, 8, 999, -3.0, 0.001, 1, 0.5, 0.5, 0, 0)
    assert lib.rk_db_create_synth(C.byref(bad), C.byref(h)) == _lib.RK_ERR_INVALID
    bad = _lib.rk_synth_desc(4, 0, 8, 0, -3.0, 0.001, 1, 0.5, 5.0, 0, 0)
    assert lib.rk_db_create_synth(C.byref(bad), C.byref(h)) == _lib.RK_ERR_INVALID


def test_c5_full_size_image():
    """BASELINE config C5 at its stated size: DNA k=12, 19 999 branches, ~3.3e10 entries = a ~200 GB image in one GPU's HBM.
    Reads are placed against it and compared with the oracle on a database rebuilt host-side from exactly the rows they touch
    (regenerated by the numpy twin of the generator); rows are also read back through rk_db_fetch_row."""
    spec = synth.make_spec("C5")
    db = ra.PhyloKmerDB.synthetic(spec)
    try:
        assert db.info.rows_bytes > 190e9 and db.info.n_entries > 3.2e10, (db.info.rows_bytes, db.info.n_entries)
        assert "place_wg_kernel" in db.kernel_name() and "OFF64" in db.kernel_name()
        seq, off = synth.make_reads(4, 48, 250, seed=1)
        dense = np.unique(synth.codes_of_reads(4, 12, seq, off))
        sdb = spec.subset_db(dense)
        odb = O.OracleDB.from_synth(sdb)
        got = ra.PlacementProcess(db).processQueries(seq, off)
        ref = odb.place(seq, off)
        st = compare_with_oracle(got, ref, odb, seq, off)
        assert st["placed"] == 48
        # rows at the far end of the blob (64-bit offsets) and a few of the touched ones
        far = np.arange(spec.space - 40, spec.space, dtype=np.uint64)
        check_rows(db, spec, np.concatenate([far, dense[:10], dense[-10:]]))
    finally:
        db.close()
