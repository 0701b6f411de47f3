"""CPU: the DB-construction oracle (oracle/rappas_build_oracle.c) against hand-derived known answers and against the
independent Python restatement (tests/pyref_build.py).  SURVEY.md section 8(f) row N4; parity unpinned (no reference outputs)."""
import numpy as np
import pytest

from oracle import oracle as O
from rappas_amd import synth
from tests import pyref_build as PB

f32 = np.float32


def _lists(off, lens):
    return [list(map(int, lens[int(off[i]):int(off[i + 1])])) for i in range(len(off) - 1)]


def _same(ref_sink, got):
    codes, off, br, sc = PB.to_csr(ref_sink)
    assert np.array_equal(got["key_codes"], codes) and np.array_equal(got["row_offsets"], off)
    assert np.array_equal(got["branch_ids"], br)
    assert np.array_equal(got["scores"].view(np.uint32), sc.view(np.uint32))


def test_hand_derived_two_sites():
    """k=2, one node, 3 sites, 2 states.  T = -1.0.
    site 0: A -0.1 / T -0.8; site 1: C -0.2 / G -0.7 (states 2,3); site 2: A -0.05 / T -1.0.
    pos 0: j=0: sum=-0.1; child (1,0): -0.1+-0.2 = -0.3 >= T -> word [0,2]; child (1,1): -0.8 -> word [0,3];
           j=1: sum=-0.8 (after the float round trip); (1,0): -1.0: is fl(fl(-0.8)+(-0.2)) < -1.0f ?  computed below, not assumed.
    Every number the assertion uses is produced by replaying the reference's statements in float32 by hand here."""
    states = np.array([[[0, 1], [2, 3], [0, 1]]], np.uint8)
    pp = np.array([[[-0.1, -0.8], [-0.2, -0.7], [-0.05, -1.0]]], np.float32)
    T = f32(-1.0)
    got = O.build_db(4, 2, states, pp, np.array([5], np.uint16), T)
    # replay (Java: float += double, one rounding)
    add = lambda s, p: f32(float(s) + float(p))
    sub = lambda s, p: f32(float(s) - float(p))
    exp = {}
    visits = 0
    for pos in range(3):  # n_sites - k + 2 = 3 explorers; the last one cannot complete a word
        s = f32(0.0)
        for j in range(2):
            visits += 1
            s = add(s, pp[0, pos, j])
            bound_here = s < T
            brk = False
            for j2 in range(2):
                if brk:
                    break
                if pos + 1 > 2:
                    continue
                visits += 1
                s = add(s, pp[0, pos + 1, j2])
                if not (s < T):
                    code = int(states[0, pos, j]) | (int(states[0, pos + 1, j2]) << 2)
                    exp[(code, 5)] = max(exp.get((code, 5), f32(-np.inf)), s)
                else:
                    brk = True  # bound reached at depth current_k + 1: the sibling loop stops
                s = sub(s, pp[0, pos + 1, j2])
            s = sub(s, pp[0, pos, j])
    _same(exp, got)
    assert got["visits"] == visits
    assert (0 | 2 << 2, 5) in exp and exp[(0 | 2 << 2, 5)] == add(f32(-0.1), f32(-0.2))


def test_running_sum_drift_is_reproduced():
    """The registered score is the explorer's RUNNING float, not a fresh left-to-right sum: values chosen so that
    fl(fl(s + a) - a) != s, which shifts the scores of later words of the same explorer by one ulp."""
    a, b = f32(-0.3), f32(-1e-8)
    pp = np.array([[[a, f32(-0.30000004)], [b, f32(-2.0)]]], np.float32)
    states = np.array([[[0, 1], [0, 1]]], np.uint8)
    sink, _, _ = PB.build(4, 2, states, pp, [0], f32(-5.0))
    got = O.build_db(4, 2, states, pp, np.array([0], np.uint16), f32(-5.0))
    _same(sink, got)
    fresh = f32(float(pp[0, 0, 1]) + float(pp[0, 1, 0]))  # what a drift-free evaluation of word [1,0] would give
    running = sink[(1, 0)]
    assert got["scores"][list(got["key_codes"]).index(1)] == running
    # (the two may or may not differ for these particular numbers; the invariant is equality with the running float)
    assert abs(float(fresh) - float(running)) <= 1e-6


@pytest.mark.parametrize("alphabet,k,n_nodes,n_sites,seed", [(4, 4, 6, 30, 1), (4, 6, 4, 24, 2), (20, 3, 3, 16, 3), (4, 8, 2, 20, 4)])
def test_c_oracle_matches_python_restatement(alphabet, k, n_nodes, n_sites, seed):
    states, pp, nb = synth.make_pp_tables(alphabet, n_nodes, n_sites, seed=seed)
    _, T = synth.thresholds(1.5, alphabet, k)
    sink, tuples, visits = PB.build(alphabet, k, states, pp, nb, T)
    got = O.build_db(alphabet, k, states, pp, nb, T)
    _same(sink, got)
    assert got["tuples"] == tuples and got["visits"] == visits
    assert len(got["key_codes"]) > 10


@pytest.mark.parametrize("limit1", [True, False])
def test_gap_jumps(limit1):
    rows = ["AC--GTAC-GTACGTTA", "ACGTGT-C-GTAC--TA", "A---GTACGGTACGTTA"]
    off, lens = synth.gap_intervals(rows)
    assert _lists(off, lens)[2] == [2] and _lists(off, lens)[1] == [3] and _lists(off, lens)[6] == [1] and _lists(off, lens)[8] == [1]
    assert _lists(off, lens)[13] == [2]
    states, pp, nb = synth.make_pp_tables(4, 4, len(rows[0]), seed=7)
    _, T = synth.thresholds(1.5, 4, 4)
    sink, tuples, visits = PB.build(4, 4, states, pp, nb, T, gaps=_lists(off, lens), limit1=limit1)
    got = O.build_db(4, 4, states, pp, nb, T, gap_off=off, gap_len=lens, limit_to_1_jump=limit1)
    _same(sink, got)
    assert got["tuples"] == tuples and got["visits"] == visits
    nojump, t0, _ = PB.build(4, 4, states, pp, nb, T)
    assert tuples > t0  # the jumps add words


def test_trailing_gap_run_is_not_an_interval():
    off, lens = synth.gap_intervals(["ACGT--", "AC-T-A"])
    assert _lists(off, lens) == [[], [], [1], [], [1], []]


from tests import golden_util as GU  # noqa: E402


@pytest.mark.parametrize("path", GU.build_cases(), ids=lambda p: p.split("/")[-1][:-5])
def test_oracle_matches_committed_vectors(path):
    args, kw, exp = GU.load_build(path)
    got = O.build_db(*args, **kw)
    assert np.array_equal(got["key_codes"], exp["key_codes"]) and np.array_equal(got["row_offsets"], exp["row_offsets"])
    assert np.array_equal(got["branch_ids"], exp["branch_ids"])
    assert np.array_equal(got["scores"].view(np.uint32), exp["score_bits"])
    assert got["tuples"] == exp["tuples"] and got["visits"] == exp["visits"]


def test_committed_vectors_exist():
    assert len(GU.build_cases()) >= 5
