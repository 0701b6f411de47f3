"""CPU: the C oracle against (a) the hand-derived golden vectors and (b) the independent Python restatement."""
import numpy as np
import pytest

from oracle import oracle as O
from rappas_amd import synth
from tests import golden_util as GU
from tests import pyref


def test_thresholds_kat():
    # float32 thresholds for the BASELINE configs (Main_DBBUILD_3.java:165-166)
    for (ns, k), want in {(4, 8): -3.4077499, (4, 10): -4.2596874, (20, 5): -5.6246934, (4, 12): -5.1116247}.items():
        p, t = O.thresholds(1.5, ns, k)
        assert t == np.float32(want)
        p2, t2 = pyref.thresholds(1.5, ns, k)
        assert (p, t) == (p2, t2)
        p3, t3 = synth.thresholds(1.5, ns, k)
        assert (p, t) == (p3, t3)
    p, _ = O.thresholds(1.5, 4, 10)
    assert p == np.float32(0.375 ** 10)


def test_compress_mer_kat():
    # DNAStatesShifted.java:115-143: ACGTACGTAC -> states 0,2,3,1,0,2,3,1,0,2 -> bytes 78 78 08 -> code 0x087878
    st = [0, 2, 3, 1, 0, 2, 3, 1, 0, 2]
    assert O.compress_mer_dna(st) == [0x78, 0x78, 0x08]
    assert O.kmer_code(4, st) == 0x087878
    assert O.compress_mer_dna(st[:8]) == [0x78, 0x78]              # k=8: whole bytes
    assert O.compress_mer_dna(st + [3, 1]) == [0x78, 0x78, 0x78]   # k=12
    assert O.compress_mer_dna([3]) == [0x03]
    assert O.compress_mer_dna([1, 1, 1, 1, 1]) == [0x55, 0x01]
    for k in range(1, 16):
        s = np.random.default_rng(k).integers(0, 4, k).tolist()
        assert O.kmer_code(4, s) == sum(v << (2 * i) for i, v in enumerate(s)) == pyref.kmer_code(4, s)
    assert O.kmer_code(20, [19, 0, 7]) == 19 | (7 << 10) == pyref.kmer_code(20, [19, 0, 7])


def test_char_tables():
    # DNA states A=0 T/U=1 C=2 G=3, both cases (DNAStatesShifted.java:182-209)
    for ch, s in zip("ATUCGatucg", [0, 1, 1, 2, 3] * 2):
        assert O.char_code(4, ch) == s
    order = {"R": [0, 3], "Y": [2, 1], "S": [2, 3], "W": [0, 1], "K": [3, 1], "M": [0, 2], "B": [2, 3, 1],
             "D": [0, 3, 1], "H": [0, 2, 1], "V": [0, 2, 3], "N": [0, 2, 3, 1]}
    for ch, alts in order.items():
        for c in (ch, ch.lower()):
            code = O.char_code(4, c)
            assert code & 0x80 and code != 0xFF
            assert O.amb_alternatives(4, code & 0x7F) == alts
    for c in ".-":
        assert O.amb_alternatives(4, O.char_code(4, c) & 0x7F) == [0, 0, 0, 0]   # never filled (:57-58)
    for c in "XZ@ 1*":
        assert O.char_code(4, c) == 0xFF
    # AA (AAStates.java:23-28, 97-123)
    for i, ch in enumerate("RHKDESTNQCGPAILMFWYV"):
        assert O.char_code(20, ch) == i == O.char_code(20, ch.lower())
    for c in "-*!Xx":
        assert O.amb_alternatives(20, O.char_code(20, c) & 0x7F) == list(range(20))
    assert O.amb_alternatives(20, O.char_code(20, "B") & 0x7F) == [3, 7]
    assert O.amb_alternatives(20, O.char_code(20, "z") & 0x7F) == [4, 8]
    assert O.amb_alternatives(20, O.char_code(20, "J") & 0x7F) == [13, 14]
    assert O.char_code(20, "U") == 0xFF and O.char_code(20, "O") == 0xFF
    assert O.char_code(20, "U", convert_uo=True) == 9 and O.char_code(20, "o", convert_uo=True) == 14
    # every character agrees with the independent table
    for alphabet in (4, 20):
        for c in range(256):
            kind, val = pyref.classify(alphabet, chr(c))
            code = O.char_code(alphabet, c)
            if kind == "bad":
                assert code == 0xFF, (alphabet, c)
            elif kind == "state":
                assert code == val
            else:
                assert code & 0x80 and O.amb_alternatives(alphabet, code & 0x7F) == val
    assert O.load().ro_max_ambig_per_mer(10, 4) == 1 and O.load().ro_max_ambig_per_mer(5, 20) == 1
    assert O.load().ro_max_ambig_per_mer(15, 4) == 1 and O.load().ro_max_ambig_per_mer(16, 4) == 2


def test_hand_computed_scores():
    """A 2-k-mer read written out by hand: S = fl(fl(Q*T) + fl(v1-T)) then + fl(v2-T)."""
    f32 = np.float32
    P, T = pyref.thresholds(1.5, 4, 4)
    v1, v2, v3 = f32(-0.25), f32(-1.5), f32(-0.75)
    rows = {pyref.kmer_code(4, [0, 2, 3, 1]): [(1, v1), (2, v3)], pyref.kmer_code(4, [2, 3, 1, 0]): [(1, v2)]}
    codes, off, br, sc = pyref.db_to_csr(dict(rows=rows))
    odb = O.OracleDB(4, 4, 3, T, P, codes, off, br, sc)
    seq = np.frombuffer(b"ACGTA", np.uint8)
    r = odb.place(seq, np.array([0, 5], np.uint64))
    QT = f32(2) * T
    s1 = f32(f32(QT + f32(v1 - T)) + f32(v2 - T))
    s2 = f32(QT + f32(v3 - T))
    assert r["n_rows"][0] == 2 and r["branch"][0, :2].tolist() == [1, 2]
    assert r["score"][0, 0] == s1 and r["score"][0, 1] == s2
    # lowest > -308 => shift 0: lwr = 10^s / (10^s1 + 10^s2), summed in heap-array order (:418)
    tot = 10.0 ** float(s2) + 10.0 ** float(s1)
    np.testing.assert_allclose(r["lwr"][0, :2], [10.0 ** float(s1) / tot, 10.0 ** float(s2) / tot], rtol=1e-15)
    assert int(r["entries"][0]) == 3


@pytest.mark.parametrize("path", GU.cases(), ids=lambda p: p.split("/")[-1][:-5])
def test_oracle_matches_golden(path):
    g = GU.load(path)
    codes, off, br, sc = g["csr"]
    odb = O.OracleDB(g["alphabet"], g["k"], g["n_branches"], g["T"], g["P"], codes, off, br, sc)
    for run in g["runs"]:
        p = run["params"]
        K = p["keep_at_most"]
        seq, roff = GU.reads_of(run)
        r = odb.place(seq, roff, keep_at_most=K, keep_factor=p["keep_factor"], amb_mode=GU.AMB[p["amb_mode"]],
                      ns_bound=p.get("ns_bound", float("-inf")))
        n_rows, branch, score, lwr, flags = GU.expected_arrays(run, K)
        assert (r["flags"] & ~np.uint32(O.RO_FLAG_TIE) == flags).all(), (p, r["flags"], flags)
        assert (r["n_rows"] == n_rows).all(), p
        assert (r["branch"] == branch).all(), p
        assert (r["score"].view(np.uint32) == score.view(np.uint32)).all(), p
        np.testing.assert_allclose(r["lwr"], lwr, rtol=1e-14, atol=0)
        assert r["entries"].tolist() == [e["H"] for e in run["expected"]]
        for i, e in enumerate(run["expected"]):
            if r["flags"][i] & O.RO_FLAG_TIE:
                assert GU.has_tie(e)
            if e["L"]:
                S, L, _ = odb.score_vector(e["read"].encode(), GU.AMB[p["amb_mode"]])
                assert L.tolist() == e["L"]
                assert {str(x): int(np.float32(S[x]).view(np.uint32)) for x in L} == e["S"]


@pytest.mark.parametrize("alphabet,k,nb", [(4, 5, 40), (4, 8, 99), (20, 3, 30)])
@pytest.mark.parametrize("amb", ["mean", "max", "skip"])
def test_oracle_matches_python_restatement_random(alphabet, k, nb, amb):
    """Random small DBs / reads incl. ambiguity, ties, short and bad reads: C oracle == tests/pyref.py bit for bit."""
    sdb = synth.make_db(alphabet, k, nb, min(alphabet ** k // 2, 400), 3000, seed=7 + k)
    rows = {}
    for r in range(sdb.n_keys):
        a, b = int(sdb.row_offsets[r]), int(sdb.row_offsets[r + 1])
        rows[int(sdb.key_codes[r])] = [(int(x), np.float32(v)) for x, v in zip(sdb.branch_ids[a:b], sdb.scores[a:b])]
    db = dict(alphabet=alphabet, k=k, n_branches=nb, T=sdb.thr_log10, P=sdb.thr, rows=rows)
    seq, off = synth.make_reads(alphabet, 60, 40, seed=3, amb_rate=0.03, bad_rate=0.05, var_len=38)
    odb = O.OracleDB.from_synth(sdb)
    K = 5
    r = odb.place(seq, off, keep_at_most=K, keep_factor=0.05, amb_mode=GU.AMB[amb])
    for i in range(len(off) - 1):
        rd = bytes(seq[int(off[i]):int(off[i + 1])]).decode()
        e = pyref.place_read(db, rd, keep_at_most=K, keep_factor=0.05, amb_mode=amb)
        want_flags = sum(GU.FLAG_BITS[f] for f in e["flags"])
        assert int(r["flags"][i]) & ~O.RO_FLAG_TIE == want_flags, (i, rd)
        assert int(r["n_rows"][i]) == len(e["rows"]), (i, rd)
        for j, (b, s, w) in enumerate(e["rows"]):
            assert int(r["branch"][i, j]) == b and r["score"][i, j].view(np.uint32) == np.float32(s).view(np.uint32), (i, j, rd)
            assert abs(r["lwr"][i, j] - w) <= 1e-14 * abs(w)
        assert int(r["entries"][i]) == e["H"]


def test_oracle_tie_flag_and_counters():
    g = GU.load([p for p in GU.cases() if "toy_dna_k4" in p][0])
    codes, off, br, sc = g["csr"]
    odb = O.OracleDB(4, g["k"], g["n_branches"], g["T"], g["P"], codes, off, br, sc)
    seq = np.frombuffer(b"ACGTACGTAC" + b"GGGGGGGG", np.uint8)
    r = odb.place(seq, np.array([0, 10, 18], np.uint64))
    assert r["flags"][0] & O.RO_FLAG_TIE            # branches 1 and 2 end on the same float (golden read 0)
    assert r["flags"][1] == 0 and r["n_rows"][1] == 0
    c = r["counters"]
    assert (c["reads"], c["placed"], c["unplaced"], c["kmers"]) == (2, 1, 1, 7 + 5)


# ---- DNA k >= 16: two ambiguity codes per k-mer are allowed (maxAmbigPerMer = floor(k^(1/4)) = 2, AmbigSequenceKnife.java:95) ----
def test_two_ambiguities_follow_the_reference_enumeration_not_the_cartesian_product():
    """AmbigSequenceKnife.java:235-260: with alternatives R = {A, G} and Y = {C, T} in one 16-mer the four words are
    (A,C) (G,T) (A,C) (G,T): word `jump` takes alt[jump % len] at each position.  (A,T) and (G,C) are never looked up, and the
    two words that are looked up count twice.  Expected score written out in float32 by hand (PlacementProcess.java:1129-1174)."""
    f32 = np.float32
    k, nb = 16, 5
    P, T = synth.thresholds(1.5, 4, k)
    base = "ACGTTGCAAGGCTTAC"
    read = base[:3] + "R" + base[4:9] + "Y" + base[10:]          # R at window position 3, Y at position 9
    code = lambda w: sum("ATCG".index(c) << (2 * i) for i, c in enumerate(w))
    word = lambda a, b: base[:3] + a + base[4:9] + b + base[10:]
    v_ac, v_gt, v_at = f32(-1.5), f32(-0.25), f32(-0.0625)
    codes = np.array([code(word("A", "C")), code(word("G", "T")), code(word("A", "T"))], np.uint64)
    odb = O.OracleDB(4, k, nb, T, P, codes, np.array([0, 1, 2, 3], np.uint64), np.array([1, 1, 2], np.uint16), np.array([v_ac, v_gt, v_at], np.float32))
    seq = np.frombuffer(read.encode(), np.uint8)
    r = odb.place(seq, np.array([0, 16], np.uint64), keep_at_most=3)
    # mean: S_amb is a float that receives double additions, in word order (A,C) (G,T) (A,C) (G,T); C_amb = 4 = W
    s_amb = f32(0.0)
    for v in (v_ac, v_gt, v_ac, v_gt):
        s_amb = f32(float(s_amb) + 10.0 ** float(v))
    avg = f32(f32(s_amb + f32(f32(4 - 4) * P)) / f32(4))
    import math
    want = f32(float(f32(f32(1) * T)) + (math.log10(float(avg)) - float(T)))   # Q = 1
    assert int(r["n_rows"][0]) == 1 and int(r["branch"][0, 0]) == 1             # branch 2 ((A,T) only) is never touched
    assert r["score"][0, 0].view(np.uint32) == want.view(np.uint32)
    assert int(r["entries"][0]) == 4
    # max: the larger of the two scores
    r = odb.place(seq, np.array([0, 16], np.uint64), keep_at_most=3, amb_mode=O.AMB_MAX)
    assert r["score"][0, 0].view(np.uint32) == f32(f32(f32(1) * T) + f32(v_gt - T)).view(np.uint32)
    # coprime counts do give the cartesian product: R = {A, G} x B = {C, G, T} -> 6 distinct words
    read2 = base[:3] + "R" + base[4:9] + "B" + base[10:]
    r = odb.place(np.frombuffer(read2.encode(), np.uint8), np.array([0, 16], np.uint64), keep_at_most=3)
    assert sorted(r["branch"][0, :2].tolist()) == [1, 2] and int(r["entries"][0]) == 3


@pytest.mark.parametrize("k", [16, 17, 20])
@pytest.mark.parametrize("amb", ["mean", "max", "skip"])
def test_oracle_matches_python_restatement_two_ambiguities(k, amb):
    sdb, genome = synth.make_motif_db(k, 60, genome_len=300, seed=k)
    rows = {}
    for r in range(sdb.n_keys):
        a, b = int(sdb.row_offsets[r]), int(sdb.row_offsets[r + 1])
        rows[int(sdb.key_codes[r])] = [(int(x), np.float32(v)) for x, v in zip(sdb.branch_ids[a:b], sdb.scores[a:b])]
    db = dict(alphabet=4, k=k, n_branches=60, T=sdb.thr_log10, P=sdb.thr, rows=rows)
    seq, off = synth.make_motif_reads(genome, 50, 60, seed=2, amb_rate=0.06, var_len=30)
    odb = O.OracleDB.from_synth(sdb)
    r = odb.place(seq, off, keep_at_most=5, keep_factor=0.05, amb_mode=GU.AMB[amb])
    two = 0
    for i in range(len(off) - 1):
        rd = bytes(seq[int(off[i]):int(off[i + 1])]).decode()
        e = pyref.place_read(db, rd, keep_at_most=5, keep_factor=0.05, amb_mode=amb)
        assert int(r["flags"][i]) & ~O.RO_FLAG_TIE == sum(GU.FLAG_BITS[f] for f in e["flags"]), (i, rd)
        assert int(r["n_rows"][i]) == len(e["rows"]), (i, rd)
        for j, (b, sc, w) in enumerate(e["rows"]):
            assert int(r["branch"][i, j]) == b and r["score"][i, j].view(np.uint32) == np.float32(sc).view(np.uint32), (i, j, rd)
            assert abs(r["lwr"][i, j] - w) <= 1e-14 * abs(w)
        assert int(r["entries"][i]) == e["H"]
        amb_pos = [p for p, c in enumerate(rd) if c not in "ATCG"]
        two += any(0 < q - p < k for p, q in zip(amb_pos, amb_pos[1:]))
    assert two > 5  # reads with two ambiguity codes inside one window were part of it
