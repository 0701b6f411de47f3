"""GPU: phylo-kmer database construction (rk_build_db) against the CPU oracle -- bit-exact CSR, same counters.
SURVEY.md section 8(f) row N4."""
import numpy as np
import pytest

import rappas_amd as ra
from oracle import oracle as O
from rappas_amd import synth

pytestmark = pytest.mark.gpu


def _check(alphabet, k, states, pp, nb, T, **kw):
    ref = O.build_db(alphabet, k, states, pp, nb, T, **kw)
    got = ra.build_db(alphabet, k, states, pp, nb, T, **kw)
    assert got.tuples == ref["tuples"] and got.visits == ref["visits"]
    assert np.array_equal(got.key_codes, ref["key_codes"])
    assert np.array_equal(got.row_offsets, ref["row_offsets"])
    assert np.array_equal(got.branch_ids, ref["branch_ids"])
    assert np.array_equal(got.scores.view(np.uint32), ref["scores"].view(np.uint32))  # the running-float scores, bit for bit
    return got, ref


@pytest.mark.parametrize("alphabet,k,n_nodes,n_sites,seed", [
    (4, 4, 20, 120, 1), (4, 8, 40, 300, 2), (4, 10, 8, 120, 3), (20, 3, 10, 60, 4), (20, 5, 12, 80, 5), (4, 2, 3, 10, 6),
    (4, 12, 2, 40, 7),
])
def test_build_matches_oracle(alphabet, k, n_nodes, n_sites, seed):
    states, pp, nb = synth.make_pp_tables(alphabet, n_nodes, n_sites, seed=seed)
    _, T = synth.thresholds(1.5, alphabet, k)
    got, _ = _check(alphabet, k, states, pp, nb, T)
    assert got.tuples >= len(got.scores) > 0


@pytest.mark.parametrize("limit1", [True, False])
def test_build_with_gap_jumps(limit1):
    rng = np.random.default_rng(11)
    L = 90
    rows = []
    for _ in range(6):
        r = list("ACGT"[int(x)] for x in rng.integers(0, 4, L))
        for _ in range(4):
            s = int(rng.integers(1, L - 8))
            for t in range(s, s + int(rng.integers(1, 6))):
                r[t] = "-"
        rows.append("".join(r))
    off, lens = synth.gap_intervals(rows)
    assert lens.size > 5
    states, pp, nb = synth.make_pp_tables(4, 12, L, seed=12)
    _, T = synth.thresholds(1.5, 4, 6)
    got, _ = _check(4, 6, states, pp, nb, T, gap_off=off, gap_len=lens, limit_to_1_jump=limit1)
    plain = ra.build_db(4, 6, states, pp, nb, T)
    assert got.tuples > plain.tuples


def test_build_edge_cases():
    states, pp, nb = synth.make_pp_tables(4, 4, 5, seed=1)
    _, T = synth.thresholds(1.5, 4, 8)
    got = ra.build_db(4, 8, states, pp, nb, T)  # alignment shorter than k: explorers run, nothing can be registered
    assert got.tuples == 0 and len(got.key_codes) == 0 and got.row_offsets.tolist() == [0]
    _check(4, 8, states, pp, nb, T)
    _check(4, 3, states[:0], pp[:0], nb[:0], T)  # no nodes
    # a threshold nothing passes / everything passes
    states, pp, nb = synth.make_pp_tables(4, 3, 12, seed=2)
    _check(4, 3, states, pp, nb, np.float32(0.5))
    got, _ = _check(4, 3, states, pp, nb, np.float32(-1e30))
    assert len(got.key_codes) == 64  # every 3-mer
    with pytest.raises(ra.RkError, match="k=16"):
        ra.build_db(4, 16, states, pp, nb, T)
    bad = states.copy(); bad[0, 0, 0] = 7
    with pytest.raises(ra.RkError, match="not a state"):
        ra.build_db(4, 3, bad, pp, nb, T)


def test_built_db_places_reads_drawn_from_its_own_posteriors():
    """End to end: posterior tables -> rk_build_db -> rk_db_create -> reads sampled from one node's most likely states are
    placed, on the GPU, on that node's branch; and the placement of the built DB equals the oracle's on the same DB."""
    n_nodes, L, k = 30, 400, 8
    states, pp, nb = synth.make_pp_tables(4, n_nodes, L, seed=21, peaked=0.97, n_branches=n_nodes)
    nb = np.arange(n_nodes, dtype=np.uint16)
    thr, T = synth.thresholds(1.5, 4, k)
    built = ra.build_db(4, k, states, pp, nb, T)
    assert len(built.key_codes) > 1000
    db = ra.PhyloKmerDB(4, k, n_nodes, T, thr, built.key_codes, built.row_offsets, built.branch_ids, built.scores)
    odb = O.OracleDB(4, k, n_nodes, T, thr, built.key_codes, built.row_offsets, built.branch_ids, built.scores)
    rng = np.random.default_rng(3)
    letters = np.frombuffer(b"ATCG", np.uint8)  # state order of the reference: A=0 T=1 C=2 G=3
    reads, truth = [], []
    for _ in range(300):
        node = int(rng.integers(0, n_nodes))
        s = int(rng.integers(0, L - 150))
        reads.append(letters[states[node, s:s + 150, 0]].tobytes())
        truth.append(node)
    seq = np.frombuffer(b"".join(reads), np.uint8)
    off = np.arange(0, 150 * 301, 150, dtype=np.uint64)
    got = ra.PlacementProcess(db).processQueries(seq, off)
    ref = odb.place(seq, off)
    from tests.util import compare_with_oracle
    compare_with_oracle(got, ref, odb, seq, off)
    hit = (got.branch[:, 0] == np.array(truth)).mean()
    assert hit > 0.95, hit
    db.close()


def test_node_batches_fold_into_the_same_database(monkeypatch, dev_lib):
    """Large inputs are explored in node batches whose reduced (k-mer, branch) -> best score sets are merged; forcing tiny
    batches must give the identical database."""
    states, pp, nb = synth.make_pp_tables(4, 14, 90, seed=9, n_branches=5)  # several nodes per branch: maxima across batches
    _, T = synth.thresholds(1.5, 4, 6)
    monkeypatch.setenv("RK_BUILD_BATCH_NODES", "3")
    got, ref = _check(4, 6, states, pp, nb, T)
    monkeypatch.delenv("RK_BUILD_BATCH_NODES")
    one = ra.build_db(4, 6, states, pp, nb, T)
    assert np.array_equal(one.scores.view(np.uint32), got.scores.view(np.uint32)) and one.tuples == got.tuples


from tests import golden_util as GU  # noqa: E402


@pytest.mark.parametrize("path", GU.build_cases(), ids=lambda p: p.split("/")[-1][:-5])
def test_build_matches_committed_vectors(path):
    args, kw, exp = GU.load_build(path)
    got = ra.build_db(*args, **kw)
    assert np.array_equal(got.key_codes, exp["key_codes"]) and np.array_equal(got.row_offsets, exp["row_offsets"])
    assert np.array_equal(got.branch_ids, exp["branch_ids"])
    assert np.array_equal(got.scores.view(np.uint32), exp["score_bits"])
    assert got.tuples == exp["tuples"] and got.visits == exp["visits"]


@pytest.mark.parametrize("seed", range(20))
def test_randomised_tables(seed):
    """Differential sweep of rk_build_db against its oracle: random alphabet / k / table shape / state counts / thresholds /
    gap intervals / jump mode (small sizes, seeded)."""
    rng = np.random.default_rng(500 + seed)
    alphabet = 4 if rng.random() < 0.65 else 20
    k = int(rng.integers(2, 9)) if alphabet == 4 else int(rng.integers(2, 4))
    n_nodes, n_sites = int(rng.integers(1, 12)), int(rng.integers(1, 40))
    states, pp, nb = synth.make_pp_tables(alphabet, n_nodes, n_sites, seed=seed, peaked=float(rng.choice([0.3, 0.8, 0.99])),
                                          n_branches=int(rng.integers(1, 9)))
    if rng.random() < 0.3:  # fewer ranked states than the alphabet has (the generic, non-vector tails)
        ns = int(rng.integers(1, alphabet))
        states, pp = np.ascontiguousarray(states[:, :, :ns]), np.ascontiguousarray(pp[:, :, :ns])
    omega = float(rng.choice([0.8, 1.0, 1.5, 2.5]))
    _, T = synth.thresholds(omega, alphabet, k)
    kw = {}
    if rng.random() < 0.4:
        rows = []
        for _ in range(4):
            r = ["A"] * n_sites
            for _ in range(3):
                s = int(rng.integers(0, n_sites))
                for t in range(s, min(n_sites, s + int(rng.integers(1, 5)))):
                    r[t] = "-"
            rows.append("".join(r))
        off, lens = synth.gap_intervals(rows)
        kw = dict(gap_off=off, gap_len=lens, limit_to_1_jump=bool(rng.random() < 0.5))
    _check(alphabet, k, states, pp, nb, T, **kw)
