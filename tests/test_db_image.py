"""The persisted database image (rk_db_save / rk_db_save_desc / rk_db_load / rk_db_image_info / rk_db_image_user): the reference's
SessionNext_v2.storeHash / load (src/main_v2/SessionNext_v2.java:110-207) for the lookup structure.  CPU part: files written from
CSR arrays without a device, header / size / checksum tests; GPU part: save -> load -> every row and every placement unchanged."""
import os

import numpy as np
import pytest

import rappas_amd as ra
from rappas_amd import _lib, synth

SHAPES = {  # name: (alphabet, k, n_branches, n_keys, n_entries, table_mode)
    "dense_nibbles": (4, 8, 999, 40000, 520000, ra.RK_TABLE_AUTO),
    "dense_bytes": (4, 6, 700, 2500, 2500 * 30, ra.RK_TABLE_DIRECT),
    "direct8": (4, 8, 99, 30000, 300000, ra.RK_TABLE_DIRECT8),
    "hash": (20, 5, 399, 30000, 300000, ra.RK_TABLE_HASH),
    "windowed": (4, 8, 7999, 40000, 520000, ra.RK_TABLE_AUTO),
    "windowed_hash_kernel": (4, 8, 40001, 50000, 650000, ra.RK_TABLE_AUTO),
    "large_tree": (4, 6, 19999, 3000, 3000 * 900, ra.RK_TABLE_AUTO),
    "protein_windowed": (20, 3, 3100, 6000, 60000, ra.RK_TABLE_AUTO),
}


def _save(sdb, path, mode, user=b""):
    ra.save_db_image(path, sdb.alphabet, sdb.k, sdb.n_branches, sdb.thr_log10, sdb.thr, sdb.key_codes, sdb.row_offsets, sdb.branch_ids,
                     sdb.scores, table_mode=mode, user=user)


@pytest.mark.parametrize("shape", ["dense_nibbles", "hash", "windowed", "large_tree"])
def test_image_written_without_a_device_describes_the_database(shape, tmp_path):
    alphabet, k, nb, nk, ne, mode = SHAPES[shape]
    sdb = synth.make_db(alphabet, k, nb, nk, ne, seed=3)
    path = str(tmp_path / "db.rkimg")
    tree = b"((A:0.1,B:0.2)C:0.3,D:0.4)R;\x00 anything at all \xff"
    _save(sdb, path, mode, user=tree)
    info, user = ra.db_image_info(path)
    want = ra.validate_db(sdb.alphabet, sdb.k, sdb.n_branches, sdb.thr_log10, sdb.thr, sdb.key_codes, sdb.row_offsets, sdb.branch_ids,
                          sdb.scores, table_mode=mode)
    for f in ("alphabet", "k", "n_branches", "table_mode", "n_keys", "n_entries", "table_slots", "table_bytes", "rows_bytes", "bits_per_symbol", "max_row_len"):
        assert getattr(info, f) == getattr(want, f), f
    assert np.float32(info.thr_log10) == sdb.thr_log10 and info.device == -1 and user == tree
    assert os.path.getsize(path) % 4096 == 0 and not os.path.exists(path + ".tmp")
    # the same arrays give the same file, byte for byte
    _save(sdb, path + "2", mode, user=tree)
    assert open(path, "rb").read() == open(path + "2", "rb").read()


def test_damaged_images_are_refused(tmp_path):
    sdb = synth.make_db(4, 7, 999, 9000, 90000, seed=5)
    path = str(tmp_path / "db.rkimg")
    _save(sdb, path, ra.RK_TABLE_AUTO, user=b"tree")
    good = open(path, "rb").read()
    lib = _lib.load()

    def refused(data, what):
        p = str(tmp_path / "bad.rkimg")
        open(p, "wb").write(data)
        with pytest.raises(_lib.RkError) as e:
            ra.db_image_info(p)
        assert e.value.code in (_lib.RK_ERR_IO, _lib.RK_ERR_UNSUPPORTED), (what, e.value)
        # ... and by the loader too, before it looks for a device
        h = __import__("ctypes").c_void_p()
        assert lib.rk_db_load(p.encode(), 0, __import__("ctypes").byref(h)) in (_lib.RK_ERR_IO, _lib.RK_ERR_UNSUPPORTED), what
        return str(e.value)

    assert "truncated" in refused(good[:-4096], "last page missing")
    assert "truncated" in refused(good + b"\0" * 4096, "a page too many")
    refused(good[:100], "shorter than a header")
    refused(b"", "empty")
    assert "magic" in refused(b"X" + good[1:], "magic")
    flipped = bytearray(good)
    flipped[40] ^= 1                                   # a header field (k)
    assert "header checksum" in refused(bytes(flipped), "header bit")
    for off in (4096 + 17, len(good) // 2, len(good) - 4096 + 1):   # table, rows, user blob
        flipped = bytearray(good)
        flipped[off] ^= 0x10
        assert "payload checksum" in refused(bytes(flipped), f"payload bit at {off}")
    swapped = bytearray(good)
    swapped[8:12] = (2).to_bytes(4, "little")          # another image version (header hash is checked after the version)
    assert "version" in refused(bytes(swapped), "version")
    with pytest.raises(_lib.RkError):
        ra.db_image_info(str(tmp_path / "does_not_exist"))


@pytest.mark.gpu
@pytest.mark.parametrize("shape", sorted(SHAPES))
def test_saved_and_loaded_database_is_the_same_database(shape, tmp_path):
    """rk_db_save of a handle -> rk_db_load: same info and kernel, every row read back through rk_db_fetch_row equal, placements
    bit for bit those of the original handle (and the oracle's); the file equals the one rk_db_save_desc writes without a device"""
    from oracle import oracle as O
    from tests.util import compare_with_oracle
    alphabet, k, nb, nk, ne, mode = SHAPES[shape]
    sdb = synth.make_db(alphabet, k, nb, nk, ne, seed=11)
    db = ra.PhyloKmerDB.from_synth(sdb, table_mode=mode)
    path = str(tmp_path / "db.rkimg")
    try:
        db.save(path, user=b"newick;")
        _save(sdb, path + ".host", mode, user=b"newick;")
        assert open(path, "rb").read() == open(path + ".host", "rb").read()
        db2 = ra.PhyloKmerDB.load(path)
        try:
            for f in ("alphabet", "k", "n_branches", "table_mode", "n_keys", "n_entries", "table_slots", "table_bytes", "rows_bytes", "max_row_len"):
                assert getattr(db.info, f) == getattr(db2.info, f), f
            assert db.kernel_name() == db2.kernel_name()
            rng = np.random.default_rng(1)
            present = rng.choice(sdb.n_keys, size=min(400, sdb.n_keys), replace=False)
            for i in present:
                b1, v1 = db.fetch_row(int(sdb.key_codes[i]))
                b2, v2 = db2.fetch_row(int(sdb.key_codes[i]))
                assert len(b1) and np.array_equal(b1, b2) and np.array_equal(v1.view(np.uint32), v2.view(np.uint32))
            absent = set(range(alphabet ** k if alphabet == 4 else 0)) - set(int(c) for c in sdb.key_codes)
            for code in list(absent)[:50]:
                assert len(db2.fetch_row(code)[0]) == 0
            seq, off = synth.make_reads(alphabet, 2500, 150 if alphabet == 4 else 100, seed=2, amb_rate=0.001, var_len=40)
            a = ra.PlacementProcess(db).processQueries(seq, off)
            b = ra.PlacementProcess(db2).processQueries(seq, off)
            for f in ("n_rows", "branch", "flags", "lwr"):
                assert np.array_equal(getattr(a, f), getattr(b, f)), f
            assert np.array_equal(a.score.view(np.uint32), b.score.view(np.uint32))
            odb = O.OracleDB.from_synth(sdb)
            compare_with_oracle(b, odb.place(seq, off), odb, seq, off)
            c = db2.clone()   # a loaded handle clones like any other
            try:
                cc = ra.PlacementProcess(c).processQueries(seq, off)
                assert np.array_equal(cc.branch, b.branch)
            finally:
                c.close()
        finally:
            db2.close()
    finally:
        db.close()
