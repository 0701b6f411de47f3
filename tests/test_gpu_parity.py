"""GPU (-m gpu): the HIP engine, called through the C ABI, against the CPU oracle and the golden vectors.

Bar: flags / n_rows / branches identical, scores bit-equal (float32), LWR within 1e-9 relative (north_star: 1e-5).
Reads whose top-(K+1) hold an exact float tie are compared as described in tests/util.py (the reference's order
among equal scores depends on its hash-map layout)."""
import ctypes as C
import os

import numpy as np
import pytest

import rappas_amd as ra
from rappas_amd import _lib, synth
from oracle import oracle as O
from tests import golden_util as GU
from tests.util import compare_with_oracle

pytestmark = pytest.mark.gpu

MODES = {"direct": ra.RK_TABLE_DIRECT, "direct8": ra.RK_TABLE_DIRECT8, "hash": ra.RK_TABLE_HASH}


@pytest.fixture(scope="module")
def c2_small():
    sdb = synth.make_config_db("C2", scale=0.2)
    return sdb, O.OracleDB.from_synth(sdb)


_EXTRA_SEEDS = int(os.environ.get("RK_TEST_EXTRA_SEEDS", "0"))  # soak runs: more seeds for the randomised sweeps


def run_case(sdb, odb, seq, off, table="direct", lanes=0, amb="mean", **kw):
    db = ra.PhyloKmerDB.from_synth(sdb, table_mode=MODES[table])
    try:
        db.set_lanes_per_read(lanes)
        pp = ra.PlacementProcess(db, ns_bound=kw.pop("ns_bound", float("-inf")))
        K = kw.get("keepAtMost", 7)
        got = pp.processQueries(seq, off, treatAmbiguities=amb != "skip", treatAmbiguitiesWithMax=amb == "max", **kw)
        ref = odb.place(seq, off, keep_at_most=K, keep_factor=kw.get("keepFactor", 0.01), amb_mode=GU.AMB[amb],
                        ns_bound=pp.ns_bound)
        st = compare_with_oracle(got, ref, odb, seq, off, amb_mode=GU.AMB[amb])
        assert got.counters["reads"] == len(off) - 1
        assert got.counters["placed"] == int((ref["flags"] & 1).sum())
        return got, ref, st
    finally:
        db.close()


@pytest.mark.parametrize("table", ["direct", "direct8", "hash"])
@pytest.mark.parametrize("lanes", [0, 8, 16, 32, 64])
def test_c1_full(table, lanes):
    """BASELINE config 1: DNA k=8, 99 branches, 1k x 150 bp reads."""
    sdb = synth.make_config_db("C1")
    seq, off = synth.make_reads(4, 1000, 150, seed=1)
    _, _, st = run_case(sdb, O.OracleDB.from_synth(sdb), seq, off, table, lanes)
    assert st["placed"] == 1000


@pytest.mark.parametrize("table", ["direct", "hash"])
@pytest.mark.parametrize("lanes", [0, 16, 32, 64])
def test_c2_scaled(c2_small, table, lanes):
    """BASELINE config 2 (DNA k=10, 999 branches) at 20 % DB size so the oracle finishes in seconds."""
    sdb, odb = c2_small
    seq, off = synth.make_reads(4, 4000, 150, seed=1)
    run_case(sdb, odb, seq, off, table, lanes)


@pytest.mark.parametrize("table", ["direct", "direct8", "hash"])
@pytest.mark.parametrize("lanes", [0, 8, 64])
def test_c4_protein(table, lanes):
    """BASELINE config 4: AA k=5 (5-bit packing), 399 branches, 100 aa reads."""
    sdb = synth.make_config_db("C4", scale=0.3)
    seq, off = synth.make_reads(20, 3000, 100, seed=1)
    run_case(sdb, O.OracleDB.from_synth(sdb), seq, off, table, lanes)


@pytest.mark.parametrize("table", ["direct", "hash"])
def test_c5_large_tree_scaled(table):
    """BASELINE config 5's tree and k (19 999 branches, k=12, 250 bp) with a DB one test can hold; rows ~50 entries."""
    sdb = synth.make_db(4, 12, 19999, 60_000, 3_000_000, seed=42)
    seq, off = synth.make_reads(4, 400, 250, seed=1)
    # make the reads hit: overwrite windows with DB k-mers
    rng = np.random.default_rng(3)
    letters = synth.DNA_LETTERS
    for r in range(400):
        for _ in range(12):
            code = int(sdb.key_codes[rng.integers(0, sdb.n_keys)])
            p = int(off[r]) + int(rng.integers(0, 238))
            seq[p:p + 12] = letters[[(code >> (2 * i)) & 3 for i in range(12)]]
    odb = O.OracleDB.from_synth(sdb)
    _, ref, st = run_case(sdb, odb, seq, off, table)          # workgroup-per-read kernel (indexed rows)
    assert st["placed"] > 300


@pytest.mark.parametrize("K", [1, 7, 16])
def test_large_tree_long_rows_workgroup_kernel(K):
    """C5-shaped rows (thousands of entries over 19 999 branches): every wave of the workgroup streams its branch slice."""
    sdb = synth.make_db(4, 6, 19999, 3000, 6_000_000, seed=11)          # mean row 2 000 entries
    seq, off = synth.make_reads(4, 60, 250, seed=17, var_len=200)
    got, ref, st = run_case(sdb, O.OracleDB.from_synth(sdb), seq, off, "direct", keepAtMost=K, keepFactor=0.0 if K == 16 else 0.01)
    assert st["placed"] >= 55


@pytest.mark.parametrize("n_branches,mean_row,kernel", [(9001, 400, "place_packed_kernel<G=64"), (9001, 1000, "place_packed_kernel<G=64"),
                                                         (13001, 400, "place_packed"), (13001, 1000, "place_wg_kernel"),
                                                         (15999, 400, "place_wg_kernel"), (15999, 1000, "place_wg_kernel")])
def test_long_rows_between_the_dense_and_the_workgroup_regimes(n_branches, mean_row, kernel):
    """rows of 400 / 1 000 entries on 9 001 ... 15 999 branches (bench.py --config L9k / L16k): the dense 64-lane kernel while a CU still
    holds three or four score vectors and the rows are short enough (or, where the image can carry window spans, the windowed kernel),
    the workgroup-per-read kernel beyond (rk_engine.hip: image_kind)"""
    sdb = _big_tree_db(n_branches, mean_row, seed=n_branches + mean_row)
    db = ra.PhyloKmerDB.from_synth(sdb)
    assert kernel in db.kernel_name(), db.kernel_name()
    db.close()
    seq, off = synth.make_reads(4, 400, 150, seed=mean_row, amb_rate=0.001, var_len=60)
    odb = O.OracleDB.from_synth(sdb)
    for K, amb in ((7, "mean"), (16, "skip")):
        _, _, st = run_case(sdb, odb, seq, off, "direct", 0, amb, keepAtMost=K)
        assert st["placed"] > 300


def test_large_tree_one_workgroup_per_cu():
    """30 000 branches: only one score vector fits a CU, the workgroup has 16 waves."""
    sdb = synth.make_db(4, 7, 30000, 8000, 2_000_000, seed=12)
    seq, off = synth.make_reads(4, 100, 200, seed=18)
    run_case(sdb, O.OracleDB.from_synth(sdb), seq, off, "hash")
    run_case(sdb, O.OracleDB.from_synth(sdb), seq, off, "direct")


@pytest.mark.parametrize("amb", ["mean", "max"])
def test_large_tree_ambiguity_multipass(amb):
    """19 999 branches: the ambiguity kernel's Samb/Camb windows cover the tree in several branch-range passes."""
    sdb = synth.make_db(4, 7, 19999, 12_000, 2_400_000, seed=8)          # mean row 200
    seq, off = synth.make_reads(4, 60, 120, seed=13, amb_rate=0.02)
    got, _, _ = run_case(sdb, O.OracleDB.from_synth(sdb), seq, off, amb=amb)
    assert (got.flags & ra.RK_FLAG_AMBIGUOUS).sum() > 30


@pytest.mark.parametrize("amb", ["mean", "max", "skip"])
@pytest.mark.parametrize("alphabet,cfg", [(4, "C1"), (20, "C4")])
def test_ambiguity_bad_and_ragged_reads(alphabet, cfg, amb):
    """IUPAC / X ambiguity (mean, max, --noamb), unsupported characters, lengths from 0 to full."""
    sdb = synth.make_config_db(cfg, scale=0.3 if cfg == "C4" else 1.0)
    rl = synth.CONFIGS[cfg][5]
    seq, off = synth.make_reads(alphabet, 1500, rl, seed=11, amb_rate=0.01, bad_rate=0.02, var_len=rl)
    got, ref, st = run_case(sdb, O.OracleDB.from_synth(sdb), seq, off, amb=amb)
    assert (got.flags & ra.RK_FLAG_AMBIGUOUS).any() and (got.flags & ra.RK_FLAG_BAD_CHAR).any()
    assert (got.flags & ra.RK_FLAG_TOO_SHORT).any()


def test_all_reads_ambiguous_dense():
    sdb = synth.make_config_db("C1")
    seq, off = synth.make_reads(4, 300, 150, seed=4, amb_rate=0.08)
    for amb in ("mean", "max"):
        run_case(sdb, O.OracleDB.from_synth(sdb), seq, off, amb=amb)


@pytest.mark.parametrize("path", GU.cases(), ids=lambda p: p.split("/")[-1][:-5])
def test_golden_vectors(path):
    """Hand-derived vectors (tests/golden/): same inputs through the C ABI."""
    g = GU.load(path)
    codes, off, br, sc = g["csr"]
    for table in ("direct", "direct8", "hash"):
        db = ra.PhyloKmerDB(g["alphabet"], g["k"], g["n_branches"], g["T"], g["P"], codes, off, br, sc,
                            table_mode=MODES[table])
        odb = O.OracleDB(g["alphabet"], g["k"], g["n_branches"], g["T"], g["P"], codes, off, br, sc)
        for run in g["runs"]:
            p = run["params"]
            K = p["keep_at_most"]
            seq, roff = GU.reads_of(run)
            pp = ra.PlacementProcess(db, ns_bound=p.get("ns_bound", float("-inf")))
            got = pp.processQueries(seq, roff, keepAtMost=K, keepFactor=p["keep_factor"],
                                    treatAmbiguities=p["amb_mode"] != "skip", treatAmbiguitiesWithMax=p["amb_mode"] == "max")
            n_rows, branch, score, lwr, flags = GU.expected_arrays(run, K)
            assert (got.flags == flags).all(), (p, got.flags, flags)
            assert (got.n_rows == n_rows).all(), p
            assert (got.score.view(np.uint32) == score.view(np.uint32)).all(), p
            np.testing.assert_allclose(got.lwr, lwr, rtol=1e-9, atol=0)
            for i, e in enumerate(run["expected"]):
                if (got.branch[i] != branch[i]).any():
                    assert GU.has_tie(e), (p, e["read"], got.branch[i], branch[i])   # tie policy: score desc, branch asc
                    for j in range(int(got.n_rows[i])):
                        assert e["S"][str(int(got.branch[i, j]))] == int(got.score[i, j].view(np.uint32))
        db.close()


def test_tie_rule_is_branch_ascending():
    g = GU.load([p for p in GU.cases() if "topk_dna_k5" in p][0])
    codes, off, br, sc = g["csr"]
    db = ra.PhyloKmerDB(4, g["k"], g["n_branches"], g["T"], g["P"], codes, off, br, sc)
    got = ra.PlacementProcess(db).processQueries(np.frombuffer(b"CCCCC", np.uint8), np.array([0, 5], np.uint64), keepFactor=0.0)
    assert got.n_rows[0] == 3 and got.branch[0, :3].tolist() == [2, 4, 9]     # three equal scores
    assert len(set(got.score[0, :3].tolist())) == 1
    db.close()


@pytest.mark.parametrize("K,kf", [(1, 0.01), (2, 0.5), (7, 0.0), (7, 1.0), (8, 0.01), (16, 0.001)])
def test_keep_at_most_and_keep_factor(c2_small, K, kf):
    sdb, odb = c2_small
    seq, off = synth.make_reads(4, 1500, 150, seed=2)
    got, _, _ = run_case(sdb, odb, seq, off, keepAtMost=K, keepFactor=kf)
    if kf == 0.0:
        assert (got.n_rows[got.flags & 1 == 1] == K).all()


def test_ns_bound_gate(c2_small):
    sdb, odb = c2_small
    seq, off = synth.make_reads(4, 800, 150, seed=5)
    base = odb.place(seq, off)
    bound = float(np.median(base["score"][:, 0]))          # gate roughly half of the reads
    got, ref, _ = run_case(sdb, odb, seq, off, ns_bound=bound)
    gated = (got.flags & ra.RK_FLAG_BELOW_NSBOUND) != 0
    assert gated.any() and not gated.all()
    assert (got.n_rows[gated] == 0).all()


def test_edge_lengths_and_empty_batch():
    sdb = synth.make_config_db("C1")
    odb = O.OracleDB.from_synth(sdb)
    db = ra.PhyloKmerDB.from_synth(sdb)
    pp = ra.PlacementProcess(db)
    got = pp.processQueries(np.zeros(0, np.uint8), np.zeros(1, np.uint64))
    assert got.n_rows.shape == (0,) and got.counters["reads"] == 0
    reads = [b"", b"A", b"ACGTACG", b"ACGTACGT", b"ACGTACGTA", b"N" * 20, b"ACGT" * 3 + b"!" + b"ACGT" * 3,
             b"acgu" * 10, bytes(synth.make_reads(4, 1, 3000, seed=8)[0]), bytes(synth.make_reads(4, 1, 16, seed=9)[0])]
    seq = np.frombuffer(b"".join(reads), np.uint8)
    off = np.zeros(len(reads) + 1, np.uint64)
    off[1:] = np.cumsum([len(r) for r in reads])
    got = pp.processQueries(seq, off)
    ref = odb.place(seq, off)
    compare_with_oracle(got, ref, odb, seq, off)
    assert got.flags[0] & ra.RK_FLAG_TOO_SHORT and got.flags[2] & ra.RK_FLAG_TOO_SHORT   # R=0, R=k-1
    assert not got.flags[3] & ra.RK_FLAG_TOO_SHORT                                      # R=k
    assert got.flags[6] & ra.RK_FLAG_BAD_CHAR and got.n_rows[6] == 0
    assert got.flags[8] & ra.RK_FLAG_PLACED                                             # 3000 bp: several list flushes
    db.close()


def test_long_reads_and_long_rows():
    """Rows longer than the lane group (chunked) and reads whose hits overflow the LDS hit list (flushed in order)."""
    sdb = synth.make_db(4, 8, 999, 30_000, 6_000_000, seed=3)          # mean row 200 entries
    odb = O.OracleDB.from_synth(sdb)
    seq, off = synth.make_reads(4, 300, 1200, seed=6, var_len=900)
    for lanes in (0, 8, 16, 32, 64):
        run_case(sdb, odb, seq, off, "direct", lanes)
    run_case(sdb, odb, seq, off, "hash", 0)
    run_case(sdb, odb, seq, off, "direct8", 0)


def test_select_prune_path():
    """Few lanes own all touched branches -> the candidate list overflows and is pruned in place (select_topk)."""
    rng = np.random.default_rng(12)
    nb, k = 999, 6
    keys = rng.choice(4 ** k, 3000, replace=False).astype(np.uint64)
    allowed = np.array([b for b in range(1, nb) if b % 16 < 5], np.uint16)
    rows_b, rows_s, off = [], [], [0]
    P, T = synth.thresholds(1.5, 4, k)
    for _ in keys:
        sel = rng.choice(allowed, size=int(rng.integers(50, 250)), replace=False)
        rows_b.append(np.sort(sel))
        rows_s.append((T * rng.random(len(sel), dtype=np.float32)).astype(np.float32))
        off.append(off[-1] + len(sel))
    sdb = synth.SynthDB(4, k, nb, P, T, keys, np.array(off, np.uint64), np.concatenate(rows_b).astype(np.uint16),
                        np.concatenate(rows_s))
    odb = O.OracleDB.from_synth(sdb)
    seq, roff = synth.make_reads(4, 200, 150, seed=2)
    for lanes in (16, 32, 64):
        got, ref, _ = run_case(sdb, odb, seq, roff, "direct", lanes)
    assert (got.n_rows > 0).sum() > 150


def test_db_validation_errors():
    sdb = synth.make_db(4, 6, 31, 200, 900, seed=1)
    bad = sdb.branch_ids.copy()
    bad[1] = bad[0]                                      # repeated branch inside a row
    with pytest.raises(ra.RkError):
        ra.PhyloKmerDB(4, 6, 31, sdb.thr_log10, sdb.thr, sdb.key_codes, sdb.row_offsets, bad, sdb.scores)
    bad = sdb.branch_ids.copy()
    bad[5] = 31                                          # >= n_branches
    with pytest.raises(ra.RkError):
        ra.PhyloKmerDB(4, 6, 31, sdb.thr_log10, sdb.thr, sdb.key_codes, sdb.row_offsets, bad, sdb.scores)
    dup = sdb.key_codes.copy()
    dup[3] = dup[2]
    for mode in MODES.values():
        with pytest.raises(ra.RkError):
            ra.PhyloKmerDB(4, 6, 31, sdb.thr_log10, sdb.thr, dup, sdb.row_offsets, sdb.branch_ids, sdb.scores, table_mode=mode)
    nan = sdb.scores.copy()
    nan[0] = np.nan
    with pytest.raises(ra.RkError):
        ra.PhyloKmerDB(4, 6, 31, sdb.thr_log10, sdb.thr, sdb.key_codes, sdb.row_offsets, sdb.branch_ids, nan)
    db = ra.PhyloKmerDB.from_synth(sdb)
    with pytest.raises(ra.RkError):
        ra.PlacementProcess(db).processQueries(np.frombuffer(b"ACGTACGT", np.uint8), np.array([0, 8], np.uint64), keepAtMost=17)
    db.close()


def test_device_api_pack_and_place(c2_small):
    """rk_pack_reads_device + rk_place_packed_device on HBM-resident buffers == host-buffer path == oracle."""
    import torch
    sdb, odb = c2_small
    seq, off = synth.make_reads(4, 5000, 150, seed=21, amb_rate=0.001, var_len=40)
    db = ra.PhyloKmerDB.from_synth(sdb)
    pp = ra.PlacementProcess(db)
    d_seq = torch.from_numpy(seq).cuda()
    d_off = torch.from_numpy(off.view(np.int64)).cuda()
    packed, lens, flags = pp.pack_reads(d_seq, d_off, 150)
    ref_packed, ref_lens = synth.pack_reads_numpy(4, seq, off, words_per_read=packed.shape[1])
    assert (packed.cpu().numpy().view(np.uint32) == ref_packed).all()
    assert (lens.cpu().numpy().view(np.uint32) == ref_lens).all()
    out = pp.place_packed(packed, lens=lens, flags_in=flags, seq_ascii=d_seq, seq_off=d_off)
    torch.cuda.synchronize()
    got = ra.Placements(out["n_rows"].cpu().numpy(), out["branch"].cpu().numpy().view(np.uint16), out["score"].cpu().numpy(),
                        out["lwr"].cpu().numpy(), out["flags"].cpu().numpy().view(np.uint32), {})
    compare_with_oracle(got, odb.place(seq, off), odb, seq, off)
    host = pp.processQueries(seq, off)
    for f in ("n_rows", "branch", "flags"):
        assert (getattr(host, f) == getattr(got, f)).all()
    assert (host.score.view(np.uint32) == got.score.view(np.uint32)).all() and (host.lwr == got.lwr).all()
    db.close()


@pytest.mark.parametrize("lanes", [0, 32])
@pytest.mark.parametrize("amb", ["mean", "max"])
def test_scores_below_threshold_take_the_general_first_touch(lanes, amb):
    """A database RAPPAS would not write (some scores < log10 threshold, i.e. negative increments): the engine must not
    use the max() first-touch shortcut that is valid only when every increment is >= 0."""
    import dataclasses
    sdb = synth.make_config_db("C2", scale=0.1)
    rng = np.random.default_rng(5)
    sc = sdb.scores.copy()
    low = rng.random(sc.shape[0]) < 0.3
    sc[low] = (sdb.thr_log10 * (1.0 + 3.0 * rng.random(int(low.sum()), dtype=np.float32))).astype(np.float32)
    assert (sc < sdb.thr_log10).sum() > 1000
    sdb = dataclasses.replace(sdb, scores=sc)
    seq, off = synth.make_reads(4, 3000, 150, seed=4, amb_rate=0.001)
    run_case(sdb, O.OracleDB.from_synth(sdb), seq, off, "direct", lanes, amb)


def test_place_batch_multi_equals_single_call(c2_small):
    """rk_place_batch_multi: several device handles of one database driven from one process, contiguous shards on host
    threads.  (A one-GPU box: the handles share device 0 -- the sharding, threading and result placement are what is tested.)"""
    sdb, odb = c2_small
    seq, off = synth.make_reads(4, 3001, 150, seed=12, amb_rate=0.001, var_len=100)
    dbs = [ra.PhyloKmerDB.from_synth(sdb) for _ in range(3)]
    try:
        pp = ra.PlacementProcess(dbs[0])
        one = pp.processQueries(seq, off)
        many = pp.processQueriesMulti(dbs, seq, off)
        for f in ("n_rows", "branch", "flags"):
            assert np.array_equal(getattr(one, f), getattr(many, f)), f
        assert np.array_equal(one.score.view(np.uint32), many.score.view(np.uint32)) and np.array_equal(one.lwr, many.lwr)
        assert one.counters == many.counters
        compare_with_oracle(many, odb.place(seq, off), odb, seq, off)
        few = pp.processQueriesMulti(dbs, seq[:int(off[2])], off[:3])  # fewer reads than handles: empty shards
        assert np.array_equal(few.branch, one.branch[:2])
    finally:
        for d in dbs:
            d.close()


@pytest.mark.parametrize("seed", range(40 + _EXTRA_SEEDS))
def test_randomised_configurations(seed):
    """Differential sweep: random alphabet / k / tree size / row lengths / read lengths / parameters / table flavour / lane-group
    width, engine against oracle (small sizes, seeded)."""
    rng = np.random.default_rng(1000 + seed)
    alphabet = 4 if rng.random() < 0.7 else 20
    k = int(rng.integers(3, 9)) if alphabet == 4 else int(rng.integers(2, 5))
    nb = int(rng.choice([1, 2, 7, 40, 200, 999, 2500, 5000, 9000, 20000, 45000, 65535]))
    space = alphabet ** k
    n_keys = int(min(space, rng.integers(1, 4000)))
    mean_len = float(rng.choice([1.0, 3.0, 12.0, 40.0, 150.0]))
    sdb = synth.make_db(alphabet, k, nb, n_keys, max(n_keys, int(n_keys * mean_len)), seed=seed)
    if rng.random() < 0.3:  # scores below the threshold: the general first-touch path
        sc = sdb.scores.copy()
        sc[::3] = sc[::3] + np.float32(sdb.thr_log10)
        import dataclasses
        sdb = dataclasses.replace(sdb, scores=sc)
    rl = int(rng.choice([k - 1, k, k + 1, 30, 150, 400]))
    seq, off = synth.make_reads(alphabet, int(rng.integers(1, 600)), max(1, rl), seed=seed + 7,
                                amb_rate=float(rng.choice([0.0, 0.003, 0.03])), bad_rate=float(rng.choice([0.0, 0.0, 0.002])),
                                var_len=int(rng.choice([0, 0, max(1, rl // 2)])))
    K = int(rng.choice([1, 3, 7, 7, 12, 16]))
    lanes = int(rng.choice([0, 0, 16, 32, 64])) if K <= 16 else 0
    odb = O.OracleDB.from_synth(sdb)
    med = float(np.median(odb.place(seq, off, keep_at_most=K)["score"][:, 0])) if len(off) > 1 else 0.0
    kw = dict(keepAtMost=K, keepFactor=float(rng.choice([0.0, 0.01, 0.5, 1.0])))
    if rng.random() < 0.25 and np.isfinite(med):
        kw["ns_bound"] = med
    table, amb = str(rng.choice(["direct", "direct8", "hash"])), str(rng.choice(["mean", "max", "skip"]))
    try:
        run_case(sdb, odb, seq, off, table, lanes, amb, **kw)
    except ra.RkError as e:
        # the only configurations the engine may refuse: lane groups narrower than keep_at_most, a fixed width on a large-tree image,
        # or a fixed width (= a dense kernel) on a tree whose score vector no CU holds.  The draw is not lost: the same database,
        # reads and parameters run again with the lane-group width the engine picks itself (always legal).
        assert lanes != 0 and ("lanes_per_read" in str(e) or "keep_at_most" in str(e) or "of LDS per read" in str(e)), e
        run_case(sdb, odb, seq, off, table, 0, amb, **kw)


@pytest.mark.parametrize("convert", [False, True])
def test_selenocysteine_and_pyrrolysine_follow_the_convertUO_switch(convert):
    """AAStates.java:118-123: with --convertUO a database treats U as C and O as L; without it both are unsupported characters."""
    sdb = synth.make_config_db("C4", scale=0.2)
    seq, off = synth.make_reads(20, 600, 80, seed=3)
    seq = seq.copy()
    rng = np.random.default_rng(2)
    for r in range(0, 600, 3):  # every third read gets a U or an O (either case)
        pos = int(off[r]) + int(rng.integers(0, 80))
        seq[pos] = ord("UuOo"[int(rng.integers(0, 4))])
    db = ra.PhyloKmerDB.from_synth(sdb, convert_uo=convert)
    odb = O.OracleDB.from_synth(sdb, convert_uo=convert)
    try:
        got = ra.PlacementProcess(db).processQueries(seq, off)
        ref = odb.place(seq, off)
        compare_with_oracle(got, ref, odb, seq, off)
        n_bad = int(((got.flags & ra.RK_FLAG_BAD_CHAR) != 0).sum())
        assert n_bad == (0 if convert else 200)
        # with the switch on, the reads place exactly as if C / L had been written
        if convert:
            plain = seq.copy()
            for a, b in ((ord("U"), ord("C")), (ord("u"), ord("C")), (ord("O"), ord("L")), (ord("o"), ord("L"))):
                plain[plain == a] = b
            same = ra.PlacementProcess(db).processQueries(plain, off)
            assert np.array_equal(same.branch, got.branch) and np.array_equal(same.score.view(np.uint32), got.score.view(np.uint32))
    finally:
        db.close()


def test_page_locked_caller_buffers_take_the_direct_dma_path(c2_small):
    """rk_host_alloc / rk_host_free: buffers the DMA uses directly (no staging copies); same results as with pageable arrays,
    also across several chunks (more than 2^19 reads would be needed for that: here the chunking is exercised by byte size)."""
    sdb, odb = c2_small
    seq, off = synth.make_reads(4, 5000, 150, seed=21, amb_rate=0.001)
    db = ra.PhyloKmerDB.from_synth(sdb)
    lib = _lib.load()
    n, K = len(off) - 1, 7
    sizes = dict(seq=seq.nbytes, off=off.nbytes, n_rows=n, branch=n * K * 2, score=n * K * 4, lwr=n * K * 8, flags=n * 4)
    ptr = {k: lib.rk_host_alloc(v) for k, v in sizes.items()}
    try:
        assert all(ptr.values())
        C.memmove(ptr["seq"], seq.ctypes.data, seq.nbytes)
        C.memmove(ptr["off"], off.ctypes.data, off.nbytes)
        res = _lib.rk_result(ptr["n_rows"], ptr["branch"], ptr["score"], ptr["lwr"], ptr["flags"])
        p = _lib.rk_params(K, 0.01, _lib.RK_AMB_MEAN, float("-inf"))
        ct = _lib.rk_counters()
        _lib.check(lib.rk_place_batch(db.handle, C.byref(p), n, C.c_void_p(ptr["seq"]), C.c_void_p(ptr["off"]), C.byref(res), C.byref(ct)))
        want = ra.PlacementProcess(db).processQueries(seq, off)  # pageable numpy arrays: the staged path
        take = lambda k, dt, shape: np.ctypeslib.as_array(C.cast(ptr[k], C.POINTER(dt)), shape).copy()
        assert np.array_equal(take("n_rows", C.c_uint8, (n,)), want.n_rows)
        assert np.array_equal(take("branch", C.c_uint16, (n, K)), want.branch)
        assert np.array_equal(take("score", C.c_uint32, (n, K)), want.score.view(np.uint32))
        assert np.array_equal(take("lwr", C.c_double, (n, K)), want.lwr)
        assert np.array_equal(take("flags", C.c_uint32, (n,)), want.flags)
        assert ct.reads == n and ct.placed == want.counters["placed"]
        compare_with_oracle(want, odb.place(seq, off), odb, seq, off)
    finally:
        for v in ptr.values():
            lib.rk_host_free(v)
        db.close()
    lib.rk_host_free(None)  # no-op


def test_host_path_across_chunk_boundaries():
    """rk_place_batch cuts a batch into chunks of 2^19 reads on two alternating workspaces (staged uploads / downloads): 1.3e6
    short reads = three chunks; checked against the oracle around both boundaries and at the ends, and for order independence."""
    sdb = synth.make_config_db("C1")
    odb = O.OracleDB.from_synth(sdb)
    n = 1_300_000
    seq, off = synth.make_reads(4, n, 40, seed=5, amb_rate=0.0005, var_len=12)
    db = ra.PhyloKmerDB.from_synth(sdb)
    try:
        got = ra.PlacementProcess(db).processQueries(seq, off)
        assert got.counters["reads"] == n
        for lo in (0, (1 << 19) - 300, (2 << 19) - 300, n - 600):
            hi = lo + 600
            sl = slice(lo, hi)
            part = ra.Placements(got.n_rows[sl], got.branch[sl], got.score[sl], got.lwr[sl], got.flags[sl], {})
            s0 = int(off[lo])
            sub_seq, sub_off = seq[s0:int(off[hi])], off[lo:hi + 1] - off[lo]
            compare_with_oracle(part, odb.place(sub_seq, sub_off), odb, sub_seq, sub_off)
        # the same reads in reverse order give the same rows
        order = np.arange(n - 1, -1, -1)
        lens = np.diff(off.astype(np.int64))
        roff = np.zeros(n + 1, np.uint64); roff[1:] = np.cumsum(lens[order])
        idx = np.repeat(off[:-1].astype(np.int64)[order] - roff[:-1].astype(np.int64), lens[order]) + np.arange(int(roff[-1]))
        rev = ra.PlacementProcess(db).processQueries(seq[idx], roff)
        assert np.array_equal(rev.branch[::-1], got.branch) and np.array_equal(rev.score[::-1].view(np.uint32), got.score.view(np.uint32))
        assert np.array_equal(rev.flags[::-1], got.flags)
    finally:
        db.close()


# ---- trees beyond one CU's LDS: branch-range passes (the reference's limit is the 16-bit node id, 65 534) ----
def _big_tree_db(n_branches, mean_row, seed=11):
    n_keys = 3000
    return synth.make_db(4, 6, n_branches, n_keys, int(n_keys * mean_row), seed=seed)


@pytest.mark.parametrize("mean_row", [12, 400])
@pytest.mark.parametrize("amb", ["mean", "max", "skip"])
def test_tree_at_the_reference_limit(mean_row, amb):
    """n_branches = 65 535 (ids up to 65 534, CustomHash_v4_FastUtil81.java:79,87): S is 262 KB, so the workgroup kernel runs
    two branch-range passes per read and the ambiguity kernel five score-vector windows"""
    sdb = _big_tree_db(65535, mean_row)
    assert int(sdb.branch_ids.max()) > 60000
    seq, off = synth.make_reads(4, 300, 150, seed=3, amb_rate=0.004, var_len=40)
    got, ref, st = run_case(sdb, O.OracleDB.from_synth(sdb), seq, off, "direct8", 0, amb)
    assert st["placed"] > 250
    db = ra.PhyloKmerDB.from_synth(sdb, table_mode=ra.RK_TABLE_DIRECT8)
    assert "passes=2" in db.kernel_name()
    db.close()
    if mean_row == 12:  # short rows with the compact table: the windowed kernel's 64 windows of 1 024 branches
        db = ra.PhyloKmerDB.from_synth(sdb)
        assert "windows=64 x 1024" in db.kernel_name(), db.kernel_name()
        db.close()
        run_case(sdb, O.OracleDB.from_synth(sdb), seq, off, "direct", 0, amb)


@pytest.mark.parametrize("n_branches", [39001, 50000])
def test_trees_between_one_and_two_lds(n_branches):
    sdb = _big_tree_db(n_branches, 60)
    seq, off = synth.make_reads(4, 200, 150, seed=4, amb_rate=0.002)
    _, _, st = run_case(sdb, O.OracleDB.from_synth(sdb), seq, off, "hash", 0, "mean")
    assert st["placed"] > 150


@pytest.mark.parametrize("passes", ["2", "4"])
def test_forced_passes_equal_the_single_pass_result(passes, monkeypatch, dev_lib):
    """a C5-sized tree (19 999 branches fits one pass) run with 2 and 4 forced passes: identical placements"""
    sdb = _big_tree_db(19999, 300, seed=5)
    odb = O.OracleDB.from_synth(sdb)
    seq, off = synth.make_reads(4, 300, 200, seed=6, amb_rate=0.003)
    one, _, _ = run_case(sdb, odb, seq, off, "direct8", 0, "mean")
    monkeypatch.setenv("RK_WG_PASSES", passes)
    many, _, _ = run_case(sdb, odb, seq, off, "direct8", 0, "mean")
    assert np.array_equal(one.branch, many.branch) and np.array_equal(one.score.view(np.uint32), many.score.view(np.uint32))
    assert np.array_equal(one.n_rows, many.n_rows) and np.array_equal(one.lwr, many.lwr)


def test_failed_shard_is_requeued_on_a_healthy_device(c2_small, monkeypatch, dev_lib):
    """a per-GPU failure re-queues the shard on another GPU.  Shard 1 of 3 reports a device failure on its
    first attempt (test knob); the call must still succeed, give the single-call results and name the device in rk_last_error."""
    sdb, odb = c2_small
    seq, off = synth.make_reads(4, 3001, 150, seed=8, amb_rate=0.001)
    dbs = [ra.PhyloKmerDB.from_synth(sdb) for _ in range(3)]
    try:
        pp = ra.PlacementProcess(dbs[0])
        want = pp.processQueries(seq, off)
        monkeypatch.setenv("RK_TEST_FAIL_SHARD", "1")
        got = pp.processQueriesMulti(dbs, seq, off)
        note = _lib.load().rk_last_error().decode()
        assert "shard 1 failed on device 0" in note and "injected" in note
        for f in ("n_rows", "branch", "flags", "lwr"):
            assert np.array_equal(getattr(got, f), getattr(want, f))
        assert np.array_equal(got.score.view(np.uint32), want.score.view(np.uint32))
        assert got.counters == want.counters
        monkeypatch.delenv("RK_TEST_FAIL_SHARD")
        pp.processQueriesMulti(dbs, seq, off)
        assert _lib.load().rk_last_error().decode() == ""
    finally:
        for d in dbs:
            d.close()


# ---- host-packed entry point (rk_pack_reads_host + rk_place_batch_packed) ----
@pytest.mark.parametrize("alphabet,cfg", [(4, "C1"), (20, "C4")])
def test_host_packer_equals_the_device_packer_and_numpy(alphabet, cfg):
    sdb = synth.make_config_db(cfg, scale=0.2 if cfg == "C4" else 1.0)
    seq, off = synth.make_reads(alphabet, 5000, 120, seed=12, amb_rate=0.004, bad_rate=0.01, var_len=60)
    db = ra.PhyloKmerDB.from_synth(sdb)
    try:
        pp = ra.PlacementProcess(db)
        packed, lens, flags = pp.pack_reads_host(seq, off)
        want, wlens = synth.pack_reads_numpy(alphabet, seq, off, words_per_read=packed.shape[1])
        assert np.array_equal(packed, want) and np.array_equal(lens, wlens)
        import torch
        dpk, dl, df = pp.pack_reads(torch.from_numpy(seq).cuda(), torch.from_numpy(off.astype(np.int64)).cuda(), int(lens.max()))
        assert np.array_equal(dpk.cpu().numpy().view(np.uint32), packed)
        assert np.array_equal(dl.cpu().numpy().view(np.uint32), lens) and np.array_equal(df.cpu().numpy().view(np.uint32), flags)
    finally:
        db.close()


@pytest.mark.parametrize("amb", ["mean", "skip"])
def test_packed_host_entry_equals_the_ascii_entry(c2_small, amb):
    """rk_place_batch_packed over records packed on the host == rk_place_batch over the characters (and == the oracle); three
    chunks, ambiguous and bad reads included; without the characters ambiguous reads come back unplaced with their flag"""
    sdb, odb = c2_small
    n = (1 << 20) + 4097
    seq, off = synth.make_reads(4, n, 150, seed=5, amb_rate=0.0002, bad_rate=0.0005, var_len=20)
    db = ra.PhyloKmerDB.from_synth(sdb)
    try:
        pp = ra.PlacementProcess(db)
        kw = dict(treatAmbiguities=amb != "skip")
        want = pp.processQueries(seq, off, **kw)
        packed, lens, flags = pp.pack_reads_host(seq, off)
        got = pp.processQueriesPacked(packed, lens=lens, flags=flags, seq=seq, seq_off=off, **kw)
        for f in ("n_rows", "branch", "flags", "lwr"):
            assert np.array_equal(getattr(got, f), getattr(want, f)), f
        assert np.array_equal(got.score.view(np.uint32), want.score.view(np.uint32)) and got.counters == want.counters
        sl = slice(n - 3000, n)
        ref = odb.place(seq[int(off[sl.start]):], off[sl.start:] - off[sl.start], amb_mode=GU.AMB[amb])
        part = ra.Placements(got.n_rows[sl], got.branch[sl], got.score[sl], got.lwr[sl], got.flags[sl], {})
        compare_with_oracle(part, ref, odb, seq[int(off[sl.start]):], off[sl.start:] - off[sl.start], amb_mode=GU.AMB[amb])
        # no characters handed over: reads with an ambiguity code are reported unplaced, everything else is unchanged
        bare = pp.processQueriesPacked(packed, lens=lens, flags=flags, **kw)
        is_amb = (flags & _lib.RK_FLAG_AMBIGUOUS) != 0
        assert is_amb.sum() > 100 and (bare.n_rows[is_amb] == 0).all() and ((bare.flags[is_amb] & _lib.RK_FLAG_AMBIGUOUS) != 0).all()
        assert np.array_equal(bare.branch[~is_amb], want.branch[~is_amb]) and np.array_equal(bare.n_rows[~is_amb], want.n_rows[~is_amb])
        # fixed-length records without lens / flags
        seq2, off2 = synth.make_reads(4, 50000, 150, seed=6)
        pk2, _, _ = pp.pack_reads_host(seq2, off2)
        a = pp.processQueriesPacked(pk2, fixed_len=150)
        b = pp.processQueries(seq2, off2)
        assert np.array_equal(a.branch, b.branch) and np.array_equal(a.score.view(np.uint32), b.score.view(np.uint32))
    finally:
        db.close()


# ---- mid-size trees: the windowed kernel (score vector held one window of <= 1000 branches at a time) ----
def _scatter_rows(sdb, seed):
    """rows whose branches are scattered over the whole tree instead of a run of neighbours: every row touches many windows"""
    rng = np.random.default_rng(seed)
    br = sdb.branch_ids.copy()
    for r in range(sdb.n_keys):
        a, b = int(sdb.row_offsets[r]), int(sdb.row_offsets[r + 1])
        br[a:b] = rng.choice(sdb.n_branches - 1, size=b - a, replace=False) + 1
    return synth.SynthDB(sdb.alphabet, sdb.k, sdb.n_branches, sdb.thr, sdb.thr_log10, sdb.key_codes, sdb.row_offsets, br, sdb.scores, sdb.seed)


@pytest.mark.parametrize("n_branches", [1291, 1500, 2801, 3999, 7999, 15999])
@pytest.mark.parametrize("amb", ["mean", "skip"])
def test_mid_size_trees_take_the_windowed_kernel(n_branches, amb):
    sdb = synth.make_db(4, 8, n_branches, 40000, 520000, seed=n_branches)
    odb = O.OracleDB.from_synth(sdb)
    seq, off = synth.make_reads(4, 3000, 150, seed=2, amb_rate=0.001, bad_rate=0.002, var_len=40)
    db = ra.PhyloKmerDB.from_synth(sdb)
    assert "place_packed16w_kernel" in db.kernel_name(), db.kernel_name()
    db.close()
    _, _, st = run_case(sdb, odb, seq, off, "direct", 0, amb)
    assert st["placed"] > 2500


@pytest.mark.parametrize("alphabet,k", [(4, 6), (20, 3)])
@pytest.mark.parametrize("longest", [240, 241])
def test_compact_table_in_its_half_size_and_byte_forms(alphabet, k, longest):
    """DIRECT stores 4-bit unit counts (24 k-mers per 16-byte block) while no row exceeds 15 units = 240 entries, bytes (12 per
    block) otherwise: every row read back through rk_db_fetch_row (the kernels' own lookup + decode), absent k-mers included, and
    placements against the oracle, in both forms"""
    nb = 700
    sdb = synth.make_db(alphabet, k, nb, 2500 if alphabet == 4 else 6000, 16 * (2500 if alphabet == 4 else 6000), seed=longest + alphabet)
    lens = (sdb.row_offsets[1:] - sdb.row_offsets[:-1]).astype(np.int64)
    # rebuild with row lengths capped at `longest` and one row of exactly that length
    lens = np.minimum(lens, 200)
    lens[7] = longest
    off = np.zeros(len(lens) + 1, dtype=np.uint64)
    np.cumsum(lens, out=off[1:])
    rng = np.random.default_rng(longest)
    br = np.concatenate([np.sort(rng.choice(nb - 1, size=int(n), replace=False)) + 1 for n in lens]).astype(np.uint16)
    sc = (sdb.thr_log10 * rng.random(int(off[-1]), dtype=np.float32)).astype(np.float32)
    sdb = synth.SynthDB(alphabet, k, nb, sdb.thr, sdb.thr_log10, sdb.key_codes, off, br, sc, sdb.seed)
    db = ra.PhyloKmerDB.from_synth(sdb, table_mode=ra.RK_TABLE_DIRECT)
    try:
        assert ("DIRECT4," in db.kernel_name()) == (longest <= 240), db.kernel_name()
        present = {int(c): r for r, c in enumerate(sdb.key_codes.tolist())}
        space = alphabet ** k
        for dense in list(range(0, space, 7)) + [space - 1]:
            code = int(synth.dense_to_code(alphabet, k, np.array([dense], dtype=np.uint64))[0])
            b, v = db.fetch_row(code)
            if code in present:
                a, e = int(off[present[code]]), int(off[present[code] + 1])
                order = np.argsort(b, kind="stable")
                assert np.array_equal(b[order], br[a:e]) and np.array_equal(v[order].view(np.uint32), sc[a:e].view(np.uint32)), code
            else:
                assert len(b) == 0, code
    finally:
        db.close()
    seq, off_r = synth.make_reads(alphabet, 1500, 120, seed=5, amb_rate=0.001, var_len=40)
    run_case(sdb, O.OracleDB.from_synth(sdb), seq, off_r, "direct", 0, "mean")


def _clade_db(k, n_branches, genome_len, seed, mean_row=12.0, jitter=5):
    """keys = the k-mers of a random genome; the rows of the k-mers of one 500-bp stretch cover the same few dozen branches: reads cut
    from the genome pile their hits on one neighbourhood, so the K best branches are adjacent ids (a clade)"""
    rng = np.random.default_rng(seed)
    g = rng.integers(0, 4, size=genome_len).astype(np.uint64)
    codes = np.zeros(genome_len - k + 1, dtype=np.uint64)
    for i in range(k):
        codes += g[i:genome_len - k + 1 + i] << np.uint64(2 * i)
    key_codes, pos = np.unique(codes, return_index=True)
    order = rng.permutation(len(key_codes))
    key_codes, pos = key_codes[order], pos[order]
    lens = np.minimum(rng.geometric(1.0 / mean_row, size=len(key_codes)), n_branches - 1).astype(np.int64)
    off = np.zeros(len(key_codes) + 1, dtype=np.uint64)
    np.cumsum(lens, out=off[1:])
    hi = np.maximum(1, n_branches - lens)
    block = pos // 500  # 500-bp stretches of the genome share a neighbourhood of the tree
    b0 = np.clip(1 + block * np.maximum(1, hi - 40) // (genome_len // 500 + 1) + rng.integers(-jitter, jitter + 1, size=len(pos)), 1, hi)
    total = int(off[-1])
    within = np.arange(total, dtype=np.int64) - np.repeat(off[:-1].astype(np.int64), lens)
    branch = (np.repeat(b0, lens) + within).astype(np.uint16)
    thr, thr_log10 = synth.thresholds(1.5, 4, k)
    scores = (thr_log10 * rng.random(total, dtype=np.float32)).astype(np.float32)
    return synth.SynthDB(4, k, n_branches, thr, thr_log10, key_codes, off, branch, scores, seed), "".join("ATCG"[int(b)] for b in g)


@pytest.mark.parametrize("n_branches", [999, 3999, 9001, 20001, 65535])
@pytest.mark.parametrize("K", [3, 7, 8])
def test_best_branches_that_are_neighbours(n_branches, K):
    """the K best branches of every read are adjacent ids: the stream heads of the fast select must keep them apart (and the exact
    paths behind it must agree when a stream has to drop one)"""
    sdb, genome = _clade_db(9, n_branches, 6000, seed=n_branches + K)
    odb = O.OracleDB.from_synth(sdb)
    seq, off = synth.make_motif_reads(genome, 2000, 150, seed=K, amb_rate=0.001, var_len=30)
    got, ref, st = run_case(sdb, odb, seq, off, "direct", 0, "mean", keepAtMost=K, keepFactor=0.0)
    assert st["placed"] == 2000
    br = ref["branch"][:, :K].astype(np.int64)
    full = ref["n_rows"] >= K
    assert full.sum() > 1500 and np.median((br.max(1) - br.min(1))[full]) < 4 * K  # really neighbours
    for lanes in (16, 64):
        if n_branches <= 9001:
            run_case(sdb, odb, seq, off, "direct", lanes, "mean", keepAtMost=K, keepFactor=0.0)


@pytest.mark.parametrize("n_branches", [399, 999, 3999, 9001, 20001, 65535])
def test_tiles_of_reads_that_hit_the_same_windows(n_branches, monkeypatch, dev_lib):
    """the 16-lane dense and the windowed kernels take their tiles through the order a counting sort by the reads' place in the tree
    gives (PlaceArgs::perm; batches of 32 768 reads and more in the product, every batch here): clade-shaped reads -- re-tiled -- mixed with uniform ones, ragged lengths, ambiguity
    codes and unsupported characters (reads the ASCII kernel or nobody places keep their own index too), a batch that is not a
    multiple of four, and a batch of uniform reads alone (the order is kept)"""
    monkeypatch.setenv("RK_RETILE_MIN_READS", "0")
    sdb, genome = _clade_db(9, n_branches, 6000, seed=n_branches)
    odb = O.OracleDB.from_synth(sdb)
    s1, o1 = synth.make_motif_reads(genome, 2501, 150, seed=5, amb_rate=0.001, var_len=40)
    s2, o2 = synth.make_reads(4, 700, 150, seed=6, amb_rate=0.001, bad_rate=0.01, var_len=100)
    seq = np.concatenate([s1, s2])
    off = np.concatenate([o1, o2[1:] + o1[-1]])
    perm = np.random.default_rng(n_branches).permutation(len(off) - 1)  # clade and uniform reads interleaved
    lens = np.diff(off.astype(np.int64))
    new_off = np.concatenate([[0], np.cumsum(lens[perm])]).astype(np.uint64)
    new_seq = np.concatenate([seq[int(off[i]):int(off[i + 1])] for i in perm])
    for K, amb in ((7, "mean"), (16, "skip")):
        _, _, st = run_case(sdb, odb, new_seq, new_off, "direct", 0, amb, keepAtMost=K)
        assert st["placed"] > 2400
    run_case(sdb, odb, s2, o2, "direct", 0, "mean")


def test_windowed_kernel_rows_scattered_over_all_windows():
    sdb = _scatter_rows(synth.make_db(4, 7, 5000, 12000, 150000, seed=9), seed=10)
    seq, off = synth.make_reads(4, 1500, 150, seed=3, amb_rate=0.001)
    _, _, st = run_case(sdb, O.OracleDB.from_synth(sdb), seq, off, "direct", 0, "mean")
    assert st["placed"] == 1500


def test_windowed_kernel_reads_too_big_for_the_item_list_probe_once_per_window(monkeypatch, dev_lib):
    """250-bp reads against a database where every k-mer has a row of ~40 entries: ~700 row units per read, more than the main
    list holds -> window ranges and the per-window probe with the row cursor (rows this dense normally take the dense kernels:
    RK_WINDOW_ALWAYS keeps the windowed image)"""
    monkeypatch.setenv("RK_WINDOW_ALWAYS", "1")
    sdb = synth.make_db(4, 6, 6001, 4096, 160000, seed=4)
    seq, off = synth.make_reads(4, 1200, 250, seed=8, var_len=120)
    _, _, st = run_case(sdb, O.OracleDB.from_synth(sdb), seq, off, "direct", 0, "mean")
    assert st["placed"] == 1200


@pytest.mark.parametrize("n_branches,mean_row", [(7999, 26), (3999, 30), (15999, 18)])
def test_windowed_kernel_reads_that_fit_the_item_list_in_two_halves(n_branches, mean_row, monkeypatch, dev_lib):
    """every k-mer hits a row of two units on average: ~300 row units per 150-bp read, more than the main list holds for the whole
    tree but not for half of its windows -> the tile is emitted once per half; mixed with short reads (fit whole) and 250-bp reads
    (do not fit a half either: the per-window probe)"""
    monkeypatch.setenv("RK_WINDOW_ALWAYS", "1")  # (rows this dense normally take the dense kernels)
    sdb = synth.make_db(4, 7, n_branches, 16384, 16384 * mean_row, seed=n_branches)
    odb = O.OracleDB.from_synth(sdb)
    parts = [synth.make_reads(4, 900, 150, seed=21), synth.make_reads(4, 300, 60, seed=22, var_len=30), synth.make_reads(4, 150, 250, seed=23)]
    seq = np.concatenate([p[0] for p in parts])
    lens = np.concatenate([(p[1][1:] - p[1][:-1]) for p in parts])
    order = np.random.default_rng(3).permutation(len(lens))  # tiles of mixed shapes
    starts = np.concatenate([[0], np.cumsum(lens)])[:-1]
    seq2 = np.concatenate([seq[int(starts[i]):int(starts[i] + lens[i])] for i in order])
    off2 = np.zeros(len(lens) + 1, dtype=np.uint64)
    off2[1:] = np.cumsum(lens[order])
    for K in (7, 3):
        _, _, st = run_case(sdb, odb, seq2, off2, "direct", 0, "mean", keepAtMost=K)
        assert st["placed"] == len(lens)


@pytest.mark.parametrize("n_branches,length", [(7999, 420), (7999, 700), (20001, 300), (31999, 150), (31999, 1200), (3999, 330),
                                               (40001, 150), (65535, 150), (65535, 500)])
def test_windowed_kernel_long_records_and_up_to_64_windows(n_branches, length):
    """records of more than 16 words (k-mers read from memory), reads emitted in several window ranges, 6-bit window ids: every
    tree the reference accepts (65 534 branch ids) as long as its rows are short"""
    sdb = synth.make_db(4, 8, n_branches, 50000, 650000, seed=n_branches + length)
    odb = O.OracleDB.from_synth(sdb)
    seq, off = synth.make_reads(4, 500, length, seed=length, amb_rate=0.0005, bad_rate=0.002, var_len=length // 2)
    db = ra.PhyloKmerDB.from_synth(sdb)
    assert "place_packed16w_kernel" in db.kernel_name(), db.kernel_name()
    db.close()
    for K in (7, 12):
        _, _, st = run_case(sdb, odb, seq, off, "direct", 0, "mean", keepAtMost=K)
        assert st["placed"] > 400
    sc = _scatter_rows(synth.make_db(4, 7, n_branches, 12000, 150000, seed=length), seed=n_branches)  # rows over all windows: span tags "to the last"
    seq, off = synth.make_reads(4, 300, min(length, 400), seed=3)
    run_case(sc, O.OracleDB.from_synth(sc), seq, off, "direct", 0, "mean")


@pytest.mark.parametrize("n_branches", [20001, 65535])
def test_windowed_kernel_protein_on_large_trees(n_branches):
    """5-bit records, up to 64 windows, reads beyond 16 record words (103+ residues), ambiguity classes B / Z / J / X"""
    sdb = synth.make_db(20, 3, n_branches, 7000, 80000, seed=n_branches)
    odb = O.OracleDB.from_synth(sdb)
    db = ra.PhyloKmerDB.from_synth(sdb)
    assert ("place_hash64_kernel<BITS=5" in db.kernel_name() or "place_packed16s_kernel<BITS=5" in db.kernel_name()) and "place_packed16w_kernel" in db.kernel_name(), db.kernel_name()
    db.close()
    for length in (90, 240):
        seq, off = synth.make_reads(20, 700, length, seed=length, amb_rate=0.002, bad_rate=0.002, var_len=length // 2)
        for K in (7, 16):
            run_case(sdb, odb, seq, off, "direct", 0, "mean", keepAtMost=K)


@pytest.mark.parametrize("alphabet,k,n_branches,n_keys,n_entries,length", [
    (4, 8, 1291, 40000, 520000, 150), (4, 8, 2801, 40000, 520000, 150), (4, 8, 3999, 40000, 520000, 150), (4, 8, 7999, 40000, 520000, 150),
    (4, 8, 15999, 50000, 650000, 160), (4, 8, 40001, 50000, 650000, 150), (4, 8, 65535, 50000, 650000, 150), (4, 7, 9001, 12000, 400000, 120),
    (20, 3, 3100, 6000, 60000, 90), (20, 3, 20001, 7000, 80000, 100)])
def test_sorted_stream_kernel_on_every_windowed_tree(alphabet, k, n_branches, n_keys, n_entries, length, monkeypatch, dev_lib):
    """place_packed16s_kernel (the round-3 windowed kernel: row units sorted by window, one accumulate stream, touched-slot lists)
    forced onto every windowed tree (it is the default beyond 4 500 branches, and for every windowed protein tree), for keep_at_most 1 ... 16, both ambiguity modes,
    reads with ambiguity codes / unsupported characters / ragged lengths, rows scattered over all windows (tiles it hands over to
    place_packed16w_kernel), and scores below the threshold (the general first-touch path)"""
    monkeypatch.setenv("RK_WSTREAM_ALWAYS", "1")
    monkeypatch.setenv("RK_NO_HASH", "1")  # (round 4: place_hash64_kernel runs first where this kernel did)
    sdb = synth.make_db(alphabet, k, n_branches, n_keys, n_entries, seed=n_branches + 1)
    odb = O.OracleDB.from_synth(sdb)
    db = ra.PhyloKmerDB.from_synth(sdb)
    assert "place_packed16s_kernel" in db.kernel_name(), db.kernel_name()
    db.close()
    seq, off = synth.make_reads(alphabet, 3000, length, seed=n_branches, amb_rate=0.001, bad_rate=0.002, var_len=length // 2)
    for K, amb in ((7, "mean"), (1, "skip"), (3, "max"), (12, "mean"), (16, "skip")):
        _, _, st = run_case(sdb, odb, seq, off, "direct", 0, amb, keepAtMost=K)
        assert st["placed"] > 2000
    import dataclasses
    sc = sdb.scores.copy()
    sc[::3] = sc[::3] + np.float32(sdb.thr_log10)
    low = dataclasses.replace(sdb, scores=sc)
    run_case(low, O.OracleDB.from_synth(low), seq, off, "direct", 0, "mean")
    if alphabet == 4:
        scat = _scatter_rows(synth.make_db(4, 7, n_branches, 12000, 150000, seed=length), seed=n_branches)
        s2, o2 = synth.make_reads(4, 600, length, seed=3)
        run_case(scat, O.OracleDB.from_synth(scat), s2, o2, "direct", 0, "mean")


@pytest.mark.parametrize("n_branches", [399, 1500, 9001])
def test_protein_records_filled_to_the_last_word(n_branches):
    """amino-acid reads of 95 ... 102 residues: what a 16-word record can hold, and all that the 16-lane kernels' seven probe rounds
    (7 x 16 = 112 k-mers) have to cover -- dense kernel, and both windowed kernels behind the switch"""
    sdb = synth.make_db(20, 5, n_branches, 30000, 300000, seed=n_branches)
    odb = O.OracleDB.from_synth(sdb)
    db = ra.PhyloKmerDB.from_synth(sdb)
    assert "PU=7" in db.kernel_name() or "place_hash64_kernel<BITS=5" in db.kernel_name(), db.kernel_name()  # (the hash kernel probes 2 x 64 k-mers; it takes batches of 32 768 reads or more)
    db.close()
    rng = np.random.default_rng(n_branches)
    lens = rng.integers(95, 103, 1200)
    lens[:8] = [102, 102, 101, 100, 99, 98, 97, 96]
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    seq = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", np.uint8)[rng.integers(0, 20, int(off[-1]))].copy()
    for K in (7, 16):
        _, _, st = run_case(sdb, odb, seq, off, "direct", 0, "mean", keepAtMost=K)
        assert st["placed"] > 300


@pytest.mark.parametrize("alphabet,k,n_branches,n_keys,n_entries,length", [
    (4, 8, 1291, 40000, 520000, 150), (4, 8, 2801, 40000, 520000, 150), (4, 8, 3999, 40000, 520000, 150), (4, 8, 7999, 40000, 520000, 150),
    (4, 8, 15999, 50000, 650000, 160), (4, 8, 40001, 50000, 650000, 150), (4, 8, 65535, 50000, 650000, 150), (4, 7, 9001, 12000, 400000, 120),
    (4, 8, 19999, 50000, 650000, 400), (20, 3, 3100, 6000, 60000, 90), (20, 3, 20001, 7000, 80000, 100)])
def test_hash_kernel_on_every_windowed_tree(alphabet, k, n_branches, n_keys, n_entries, length, monkeypatch, dev_lib):
    """place_hash64_kernel (round 4: a read's scores in an LDS hash table keyed by branch, four row units a step, the units of a step
    one after the other whenever two of them meet in a branch) forced onto every windowed tree, for keep_at_most 1 ... 16, all
    ambiguity modes, reads with ambiguity codes / unsupported characters / ragged lengths (400-symbol reads: records of more than 16
    words, tables that overflow -> tiles handed to place_packed16w_kernel), rows scattered over the whole tree, scores below the
    threshold, and a table small enough that most reads are handed over"""
    monkeypatch.setenv("RK_HASH_ALWAYS", "1")
    sdb = synth.make_db(alphabet, k, n_branches, n_keys, n_entries, seed=n_branches + 1)
    odb = O.OracleDB.from_synth(sdb)
    db = ra.PhyloKmerDB.from_synth(sdb)
    assert "place_hash64_kernel" in db.kernel_name(), db.kernel_name()
    db.close()
    seq, off = synth.make_reads(alphabet, 3000, length, seed=n_branches, amb_rate=0.001, bad_rate=0.002, var_len=length // 2)
    for K, amb in ((7, "mean"), (1, "skip"), (3, "max"), (12, "mean"), (16, "skip")):
        _, _, st = run_case(sdb, odb, seq, off, "direct", 0, amb, keepAtMost=K)
        assert st["placed"] > 2000
    import dataclasses
    sc = sdb.scores.copy()
    sc[::3] = sc[::3] + np.float32(sdb.thr_log10)
    low = dataclasses.replace(sdb, scores=sc)
    run_case(low, O.OracleDB.from_synth(low), seq, off, "direct", 0, "mean")
    if alphabet == 4:
        scat = _scatter_rows(synth.make_db(4, 7, n_branches, 12000, 150000, seed=length), seed=n_branches)
        s2, o2 = synth.make_reads(4, 600, length, seed=3)
        run_case(scat, O.OracleDB.from_synth(scat), s2, o2, "direct", 0, "mean")
        monkeypatch.setenv("RK_HASH_KEY_SLACK", "1400")  # a table of 648 keys: most reads overflow it, some in the middle of a step
        run_case(sdb, odb, seq, off, "direct", 0, "mean")


@pytest.mark.parametrize("n_branches", [2801, 9001, 65535])
def test_hash_kernel_units_of_a_step_that_meet_in_a_branch(n_branches, monkeypatch, dev_lib):
    """clade-shaped reads: consecutive k-mers hit the same few dozen branches, so nearly every step of place_hash64_kernel holds two
    units that update the same branch and is applied unit by unit -- float32 sums in k-mer order, bit for bit; mixed with uniform
    reads (steps without a clash) in one batch, re-tiled"""
    monkeypatch.setenv("RK_HASH_ALWAYS", "1")
    monkeypatch.setenv("RK_RETILE_MIN_READS", "0")
    sdb, genome = _clade_db(9, n_branches, 6000, seed=n_branches, mean_row=20.0)
    odb = O.OracleDB.from_synth(sdb)
    s1, o1 = synth.make_motif_reads(genome, 2000, 150, seed=5, amb_rate=0.001, var_len=40)
    s2, o2 = synth.make_reads(4, 501, 150, seed=6, var_len=100)
    seq = np.concatenate([s1, s2])
    off = np.concatenate([o1, o2[1:] + o1[-1]])
    for K in (7, 16):
        _, _, st = run_case(sdb, odb, seq, off, "direct", 0, "mean", keepAtMost=K, keepFactor=0.0)
        assert st["placed"] > 1900


@pytest.mark.parametrize("rows,n_branches", [("sparse", 12001), ("sparse", 20001), ("dense", 40001), ("dense", 60001), ("sparse-small-table", 12001), ("sparse-small-table", 30001)])
def test_first_kernel_by_the_shape_of_the_batch(rows, n_branches, monkeypatch, dev_lib):
    """between the two crossing points of a database (C2-like rows: 28 000 / 56 000 branches; few row units a read, the 1 024-slot table:
    ~2 000 / 24 000) place_hash64_kernel and place_packed16s_kernel are both launched and the verdict of the re-tiling pre-pass on the
    batch -- uniform reads keep their order, reads of a clade are re-tiled -- says which of them runs: a clade-shaped batch, a uniform one
    and a mixed one all equal the oracle (the pre-pass switched on for small batches)"""
    monkeypatch.setenv("RK_RETILE_MIN_READS", "0")
    if rows.endswith("small-table"):  # (the 1 024-slot instantiation: taken by itself only when a read all of whose k-mers hit fits it)
        monkeypatch.setenv("RK_HASH_SMALL_TABLE", "1")
    if rows.startswith("sparse"):
        sdb, genome = _clade_db(9, n_branches, 6000, seed=n_branches, mean_row=12.0)
        s1, o1 = synth.make_motif_reads(genome, 2500, 150, seed=5, amb_rate=0.001, var_len=40)
    else:
        sdb, g = synth.make_clade_db(k=10, n_branches=n_branches, genome_len=400_000)
        s1, o1 = synth.make_clade_reads(g, 2500, 150)
    odb = O.OracleDB.from_synth(sdb)
    db = ra.PhyloKmerDB.from_synth(sdb)
    name = db.kernel_name()
    db.close()
    # (the name spells the plan out: the kernel of uniform batches first, then the classes of batches that take another)
    assert "place_hash64_kernel" in name and ("LOGS=10" in name) == rows.endswith("small-table"), name
    if rows == "sparse":  # (few row units a read: clade-shaped batches change over at ~19 300 branches)
        assert ("clade-shaped -> place_packed16s_kernel" if n_branches == 12001 else "clade-shaped -> place_hash64_kernel with 1 024 slots") in name, name
    if rows == "dense":
        assert "clade-shaped -> place_hash64_kernel with 1 024 slots" in name, name
    s2, o2 = synth.make_reads(4, 2500, 150, seed=6, var_len=60, amb_rate=0.001, bad_rate=0.002)
    for seq, off in ((s1, o1), (s2, o2), (np.concatenate([s1, s2]), np.concatenate([o1, o2[1:] + o1[-1]]))):
        for K in (7, 12):
            _, _, st = run_case(sdb, odb, seq, off, "direct", 0, "mean", keepAtMost=K)
            assert st["placed"] > 1000


def test_first_kernel_by_how_often_the_batch_hits(monkeypatch, dev_lib):
    """a database that holds a fifth of the k-mer codes, on 20 001 branches: random reads bring ~300 row entries -- the 1 024-slot table of
    place_hash64_kernel -- but reads cut from the sequence the keys come from hit with every k-mer, four times that.  The pre-pass counts the
    sampled k-mers that have a row and the batch goes to the small table only when that is no more than a random read's share: a sparse
    batch, a dense one (rows anywhere in the tree: no clade) and a mixed one all equal the oracle"""
    monkeypatch.setenv("RK_RETILE_MIN_READS", "0")
    sdb, g = synth.make_clade_db(k=10, n_branches=20001, genome_len=200_000, seed=9)
    rng = np.random.default_rng(4)
    lens = np.diff(sdb.row_offsets.astype(np.int64))
    b0 = rng.integers(1, np.maximum(2, sdb.n_branches - lens))  # (every row a run of branches somewhere in the tree, not in its stretch's neighbourhood)
    within = np.arange(int(sdb.row_offsets[-1]), dtype=np.int64) - np.repeat(sdb.row_offsets[:-1].astype(np.int64), lens)
    sdb = synth.SynthDB(sdb.alphabet, sdb.k, sdb.n_branches, sdb.thr, sdb.thr_log10, sdb.key_codes, sdb.row_offsets,
                        (np.repeat(b0, lens) + within).astype(np.uint16), sdb.scores, sdb.seed)
    odb = O.OracleDB.from_synth(sdb)
    s1, o1 = synth.make_clade_reads(g, 2500, 150, seed=3)          # every k-mer has a row
    s2, o2 = synth.make_reads(4, 2500, 150, seed=6, var_len=40)    # a fifth of them
    for seq, off in ((s2, o2), (s1, o1), (np.concatenate([s1, s2]), np.concatenate([o1, o2[1:] + o1[-1]]))):
        _, _, st = run_case(sdb, odb, seq, off, "direct", 0, "mean")
        assert st["placed"] > 2000


@pytest.mark.parametrize("kernel", ["hash", "sorted"])
@pytest.mark.parametrize("seed", range(6 + _EXTRA_SEEDS))
def test_short_row_kernels_of_big_trees_with_scores_from_a_handful_of_values(seed, kernel, request, monkeypatch):
    """equal sums everywhere: every score of the database is one of eight values (dyadic fractions of the threshold, so that sums of
    them collide exactly), on windowed trees of 4 501 ... 65 535 branches, keep_at_most 1 ... 16.  The stream heads of
    place_hash64_kernel (the product's choice beyond 36 000 branches; RK_HASH_ALWAYS) and of place_packed16s_kernel (RK_NO_HASH) are fed in any branch order; a tie that
    could be among the K best has to send the read through the exact ranking (or to place_packed16w_kernel) -- the result must be the
    oracle's branch for branch wherever the oracle's own order is defined (tests/util.py compares exact ties as sets)."""
    import dataclasses
    request.getfixturevalue("dev_lib")
    monkeypatch.setenv("RK_NO_HASH" if kernel == "sorted" else "RK_HASH_ALWAYS", "1")
    rng = np.random.default_rng(1000 + seed)
    nb = int(rng.choice([4501, 7000, 9001, 15999, 20001, 33001, 65535]))
    sdb = synth.make_db(4, 8, nb, 40000, 520000, seed=seed)
    q = (rng.integers(1, 9, sdb.scores.shape[0]).astype(np.float32) / np.float32(8.0)) * np.float32(sdb.thr_log10)
    sdb = dataclasses.replace(sdb, scores=q.astype(np.float32))
    db = ra.PhyloKmerDB.from_synth(sdb)
    assert ("place_hash64_kernel" if kernel == "hash" else "place_packed16s_kernel") in db.kernel_name(), db.kernel_name()
    db.close()
    seq, off = synth.make_reads(4, 2500, 150, seed=seed, var_len=70)
    K = int(rng.choice([1, 3, 7, 8, 12, 16]))
    kf = float(rng.choice([0.0, 0.01, 0.5]))
    got, _, st = run_case(sdb, O.OracleDB.from_synth(sdb), seq, off, "direct", 0, "mean", keepAtMost=K, keepFactor=kf)
    assert st["placed"] > 2000 and st["ties"] > 50, st
    # ... and the dense 64-lane kernel's, row for row: among equal scores the engine's order is the branch's, whichever kernel ran
    if nb <= 33001:  # (beyond, a whole score vector no longer fits a CU's LDS)
        dense, _, _ = run_case(sdb, O.OracleDB.from_synth(sdb), seq, off, "direct", 64, "mean", keepAtMost=K, keepFactor=kf)
        assert np.array_equal(got.n_rows, dense.n_rows) and np.array_equal(got.branch, dense.branch)
        assert np.array_equal(got.score.view(np.uint32), dense.score.view(np.uint32)) and np.array_equal(got.lwr, dense.lwr)


def test_dense_rows_on_a_mid_size_tree_take_the_dense_kernels():
    """beyond ~2.2 row units per k-mer code a read's units no longer fit the windowed kernel's lists (scripts/row_length_sweep.py)"""
    sdb = synth.make_db(4, 6, 5001, 4096, 4096 * 60, seed=2)  # every k-mer present, rows of ~60 entries: ~4 units per code
    db = ra.PhyloKmerDB.from_synth(sdb)
    assert "place_packed16w_kernel" not in db.kernel_name(), db.kernel_name()
    db.close()
    seq, off = synth.make_reads(4, 800, 150, seed=12, var_len=50)
    run_case(sdb, O.OracleDB.from_synth(sdb), seq, off, "direct", 0, "mean")


def test_windowed_kernel_protein_and_keep_at_most():
    sdb = synth.make_db(20, 3, 3100, 6000, 60000, seed=6)
    odb = O.OracleDB.from_synth(sdb)
    seq, off = synth.make_reads(20, 1500, 100, seed=1, amb_rate=0.002, var_len=30)
    for K in (1, 8, 9, 16):  # beyond 8 the windows' winners are merged through the LDS instead of lane rotations
        run_case(sdb, odb, seq, off, "direct", 0, "mean", keepAtMost=K, keepFactor=0.0001)


@pytest.mark.parametrize("K", [9, 12, 16])
@pytest.mark.parametrize("n_branches", [1500, 7999, 15999])
def test_windowed_kernel_keep_at_most_beyond_eight(n_branches, K):
    """scattered and clade-shaped reads: the fast pass, the exact pass (a stream drops candidates far more often with 16 winners) and
    the merge of two sets of up to 16 winners"""
    sdb = synth.make_db(4, 8, n_branches, 40000, 520000, seed=n_branches + K)
    seq, off = synth.make_reads(4, 1500, 150, seed=K, amb_rate=0.001, var_len=60)
    run_case(sdb, O.OracleDB.from_synth(sdb), seq, off, "direct", 0, "mean", keepAtMost=K, keepFactor=0.0)
    cdb, genome = _clade_db(9, n_branches, 6000, seed=K)
    seq, off = synth.make_motif_reads(genome, 1200, 150, seed=K, var_len=30)
    run_case(cdb, O.OracleDB.from_synth(cdb), seq, off, "direct", 0, "mean", keepAtMost=K, keepFactor=0.0)


def test_windowed_and_dense_kernels_agree():
    sdb = synth.make_db(4, 8, 4500, 40000, 520000, seed=77)
    seq, off = synth.make_reads(4, 20000, 150, seed=5)
    db = ra.PhyloKmerDB.from_synth(sdb)
    try:
        pp = ra.PlacementProcess(db)
        a = pp.processQueries(seq, off)
        db.set_lanes_per_read(32)  # an explicit lane-group width selects the dense kernel
        b = pp.processQueries(seq, off)
        assert "place_packed_kernel" in db.kernel_name()
        for f in ("n_rows", "branch", "flags", "lwr"):
            assert np.array_equal(getattr(a, f), getattr(b, f))
        assert np.array_equal(a.score.view(np.uint32), b.score.view(np.uint32))
    finally:
        db.close()


@pytest.mark.parametrize("shape", ["small", "windowed", "large_tree"])
def test_cloned_handles_place_identically(shape):
    """rk_db_clone: device-to-device copy of the image (table, rows, window spans); shards over the source and two clones give the
    single-call records"""
    sdb = {"small": lambda: synth.make_config_db("C1"), "windowed": lambda: synth.make_db(4, 8, 4001, 30000, 400000, seed=2),
           "large_tree": lambda: synth.make_db(4, 6, 20001, 3000, 600000, seed=3)}[shape]()
    seq, off = synth.make_reads(4, 3000, 150, seed=9, amb_rate=0.001)
    a = ra.PhyloKmerDB.from_synth(sdb)
    b = a.clone(0)
    c = b.clone(0)
    try:
        assert b.kernel_name() == a.kernel_name() and b.info.rows_bytes == a.info.rows_bytes and b.info.n_entries == a.info.n_entries
        pp = ra.PlacementProcess(a)
        want = pp.processQueries(seq, off)
        got = pp.processQueriesMulti([a, b, c], seq, off)
        alone = ra.PlacementProcess(c).processQueries(seq, off)
        for g in (got, alone):
            for f in ("n_rows", "branch", "flags", "lwr"):
                assert np.array_equal(getattr(g, f), getattr(want, f))
            assert np.array_equal(g.score.view(np.uint32), want.score.view(np.uint32))
        code = int(sdb.key_codes[5])
        assert all(np.array_equal(x, y) for x, y in zip(a.fetch_row(code), c.fetch_row(code)))
    finally:
        for d in (a, b, c):
            d.close()


# ---- DNA k >= 16: hashed table (4^k codes), up to two ambiguity codes per k-mer (AmbigSequenceKnife.java:95,235-260) ----
@pytest.mark.parametrize("k", [16, 17, 24, 31])
@pytest.mark.parametrize("amb", ["mean", "max", "skip"])
def test_long_kmers_and_two_ambiguity_codes_per_kmer(k, amb):
    sdb, genome = synth.make_motif_db(k, 700, genome_len=3000, seed=k)
    odb = O.OracleDB.from_synth(sdb)
    seq, off = synth.make_motif_reads(genome, 1500, 150, seed=4, amb_rate=0.012, var_len=60)
    got, ref, st = run_case(sdb, odb, seq, off, "hash", 0, amb)
    assert st["placed"] > 1400
    lens = (off[1:] - off[:-1]).astype(int)
    n_amb = np.add.reduceat((~np.isin(seq, np.frombuffer(b"ATCG", np.uint8))).astype(int), off[:-1].astype(int))
    assert (n_amb >= 2).sum() > 200  # plenty of reads with two and more ambiguity codes


# ---- amino acids k = 5..12 (5 bits a symbol: up to 60-bit codes, hashed table), with the B/Z/J/X ambiguity classes ----
@pytest.mark.parametrize("k", [5, 6, 8, 11, 12])
@pytest.mark.parametrize("amb", ["mean", "max", "skip"])
def test_long_protein_kmers(k, amb):
    sdb, genome = synth.make_motif_db(k, 900, genome_len=2500, seed=40 + k, alphabet=20)
    odb = O.OracleDB.from_synth(sdb)
    seq, off = synth.make_motif_reads(genome, 1200, 120, seed=9, amb_rate=0.01, var_len=50)
    got, ref, st = run_case(sdb, odb, seq, off, "hash", 0, amb)
    assert st["placed"] > 1000
    if k == 8:
        for lanes in (16, 32, 64):
            run_case(sdb, odb, seq, off, "hash", lanes, amb)


def test_long_protein_kmers_mid_size_and_large_trees():
    for nb, seed in ((3999, 1), (7001, 2), (20001, 3), (65535, 4)):
        sdb, genome = synth.make_motif_db(9, nb, genome_len=1500, mean_row=120, seed=seed, alphabet=20)
        odb = O.OracleDB.from_synth(sdb)
        seq, off = synth.make_motif_reads(genome, 400, 100, seed=seed, amb_rate=0.006, var_len=30)
        run_case(sdb, odb, seq, off, "hash", 0, "mean")


@pytest.mark.parametrize("seed", range(24 + _EXTRA_SEEDS))
def test_randomised_long_kmers_and_tree_sizes(seed):
    """Differential sweep over the corners the small sweep cannot reach: hashed k-mer spaces (DNA k 13..31, amino acids k 5..12),
    trees from a few branches to the reference's 65 534, mid-size (windowed) and large (indexed, multi-pass) images."""
    rng = np.random.default_rng(7000 + seed)
    alphabet = 4 if rng.random() < 0.6 else 20
    k = int(rng.integers(13, 32)) if alphabet == 4 else int(rng.integers(5, 13))
    nb = int(rng.choice([3, 250, 999, 2801, 3500, 6000, 11000, 15999, 16001, 25000, 39001, 52000, 65535]))
    mean_row = float(rng.choice([2.0, 8.0, 40.0, 150.0]))
    sdb, genome = synth.make_motif_db(k, nb, genome_len=int(rng.integers(300, 2500)), n_variants=int(rng.integers(0, 4)),
                                      mean_row=mean_row, seed=seed, alphabet=alphabet)
    odb = O.OracleDB.from_synth(sdb)
    rl = int(rng.choice([k, k + 1, 60, 150, 290]))
    seq, off = synth.make_motif_reads(genome, int(rng.integers(1, 500)), rl, seed=seed + 3,
                                      amb_rate=float(rng.choice([0.0, 0.004, 0.02])), var_len=int(rng.choice([0, rl // 3])))
    K = int(rng.choice([1, 3, 7, 7, 12, 16]))
    kw = dict(keepAtMost=K, keepFactor=float(rng.choice([0.0, 0.01, 0.5, 1.0])))
    run_case(sdb, odb, seq, off, "hash", 0, str(rng.choice(["mean", "max", "skip"])), **kw)


def test_long_kmers_large_tree_and_lane_widths():
    sdb, genome = synth.make_motif_db(18, 20001, genome_len=2000, mean_row=150, seed=3)
    odb = O.OracleDB.from_synth(sdb)
    seq, off = synth.make_motif_reads(genome, 300, 200, seed=5, amb_rate=0.008)
    run_case(sdb, odb, seq, off, "hash", 0, "mean")
    sdb2, genome2 = synth.make_motif_db(20, 300, genome_len=2000, seed=8)
    odb2 = O.OracleDB.from_synth(sdb2)
    seq2, off2 = synth.make_motif_reads(genome2, 500, 150, seed=6, amb_rate=0.01)
    for lanes in (8, 16, 32, 64):
        run_case(sdb2, odb2, seq2, off2, "hash", lanes, "mean")
