"""Rows N1-N3 of SURVEY.md section 8(f): FASTA ingest + dedup, --jsondb ingest, jplace writer (host side, no GPU compute).

Known answers are derived by hand from the reference sources cited in rappas_amd/hostio.py; number layouts are the
documented java.lang.Float/Double.toString behaviour.  (Parity of these rows is unpinned: no JVM here.)
"""
import json
import re

import numpy as np
import pytest

from rappas_amd import hostio, synth


def test_fasta_reader_semantics():
    txt = "# comment\n>r1 desc\nACGT\n\nAC-GT\n>r2\n  ACGTN  \n>empty\n>r3\nAC\n"
    rec = hostio.read_fasta(txt)
    assert rec == [("r1 desc", "ACGTAC-GT"), ("r2", "ACGTN"), ("empty", ""), ("r3", "AC")]
    assert hostio.read_fasta(b"ACGT\n>x\nAC\n") == [("x", "AC")]  # text before the first header is dropped
    assert hostio.read_fasta("") == []


def test_dedup_rules():
    rec = [("a first", "ACGT"), ("b second", "AC-GT"), ("c", "acgt"), ("d x y", "ACGT"), ("e", "TTTT")]
    uniq, names = hostio.dedup_reads(rec)
    # gaps are ignored by the checksum, case is not; first occurrence keeps the whole header, duplicates are cut at ' '
    assert uniq == [("a first", "ACGT"), ("c", "acgt"), ("e", "TTTT")]
    assert names == [["a first", "b", "d"], ["c"], ["e"]]
    seq, off = hostio.pack_batch([s for _, s in uniq])
    assert bytes(seq) == b"ACGTacgtTTTT" and off.tolist() == [0, 4, 8, 12]
    seq, off = hostio.pack_batch([])
    assert seq.size == 0 and off.tolist() == [0]


def test_notplaced_log_lists_every_occurrence_with_its_full_header():
    """PlacementProcess.java:797-806 + :1046: checksums are registered for placed reads only, so each copy of an unplaced read
    is placed again and logged again, with the header as it stands in the file"""
    rec = [("a first", "ACGT"), ("n1 some text", "NNNN"), ("b", "AC-GT"), ("n2", "NN-NN"), ("n3 x", "GGGG")]
    uniq, names = hostio.dedup_reads(rec)
    assert [h for h, _ in uniq] == ["a first", "n1 some text", "n3 x"]
    placed = np.array([True, False, False])
    assert hostio.notplaced_log(rec, uniq, placed) == "n1 some text\nn2\nn3 x\n"
    assert hostio.notplaced_log(rec, uniq, np.array([True, True, True])) == ""


def test_newick_ids_and_jplace_edges():
    t = hostio.parse_newick("((A:0.1,B:0.2)C:0.3,D:0.4)R;")
    assert [(n.id, n.label) for n in t.nodes] == [(0, "R"), (1, "C"), (2, "A"), (3, "B"), (4, "D")]
    assert {n.label: n.jplace_edge for n in t.nodes} == {"A": 0, "B": 1, "C": 2, "D": 3, "R": 4}
    assert t.rooted
    assert t.nodes[1].bl == np.float32(0.3) and t.nodes[0].bl == np.float32(0.0)
    assert t.nodes[2].parent is t.nodes[1] and t.nodes[1].parent is t.root
    assert t.jplace_newick() == ("((A:0.100000001490{0},B:0.200000002980{1})C:0.300000011921{2},D:0.400000005960{3})"
                                 "R:0.000000000000{4};")


def test_newick_unrooted_and_unlabelled():
    t = hostio.parse_newick("(A:1,(B:2.5,C:1e-3):0.5,(D:1,E:1):1234.5);")
    assert [n.label for n in t.nodes] == ["", "A", "", "B", "C", "", "D", "E"]
    assert not t.rooted
    assert [n.jplace_edge for n in t.nodes] == [7, 0, 3, 1, 2, 6, 4, 5]
    s = t.jplace_newick()
    assert s == ("(A:1.000000000000{0},(B:2.500000000000{1},C:0.001000000047{2}):0.500000000000{3},"
                 "(D:1.000000000000{4},E:1.000000000000{5}):1,234.500000000000{6});")
    assert hostio.write_newick(t, False, False, False) == "(A,(B,C),(D,E));"
    # a jplace tree string parses back to the same topology and lengths
    t2 = hostio.parse_newick(re.sub(r"\{\d+\}", "", s.replace(",234", "234")))
    assert [n.bl for n in t2.nodes] == [n.bl for n in t.nodes]


@pytest.mark.parametrize("x,expect", [
    (0.0, "0.0"), (-0.0, "-0.0"), (1.0, "1.0"), (0.5, "0.5"), (100.0, "100.0"), (1234567.0, "1234567.0"),
    (1.0e7, "1.0E7"), (1.2345e10, "1.2345E10"), (0.001, "0.001"), (0.0001, "1.0E-4"), (9.999e-4, "9.999E-4"),
    (-3.25, "-3.25"), (1e-300, "1.0E-300"), (0.1 + 0.2, "0.30000000000000004"), (123.456, "123.456"),
    (float("nan"), "null"), (float("inf"), "null"),
])
def test_java_double_to_string(x, expect):
    assert hostio.java_double_to_string(x) == expect


@pytest.mark.parametrize("x,expect", [
    (0.1, "0.1"), (-1.5, "-1.5"), (-12.345678, "-12.345678"), (1e-5, "1.0E-5"), (3.4028235e38, "3.4028235E38"),
    (16777216.0, "1.6777216E7"), (0.05, "0.05"), (-308.25, "-308.25"), (0.33333334, "0.33333334"),
])
def test_java_float_to_string(x, expect):
    assert hostio.java_float_to_string(np.float32(x)) == expect


def _toy_tree():
    return hostio.parse_newick("((A:0.1,B:0.2)C:0.3,D:0.4)R;")


def test_jplace_document_layout():
    t = _toy_tree()
    n_rows = np.array([2, 0, 1], np.uint8)
    branch = np.array([[2, 1], [0xFFFF, 0xFFFF], [4, 0xFFFF]], np.uint16)
    score = np.array([[-1.5, -2.25], [-np.inf, -np.inf], [-0.125, -np.inf]], np.float32)
    lwr = np.array([[0.75, 0.25], [0, 0], [1.0, 0]], np.float64)
    names = [["r1 full header", "r1dup"], ["r2"], ["r/3"]]
    pl = hostio.jplace_placements(t, names, n_rows, branch, score, lwr)
    assert [p[0] for p in pl] == [[["0", "-1.5", "0.75", "0.05", "0.0"], ["2", "-2.25", "0.25", "0.15", "0.0"]],
                                  [["3", "-0.125", "1.0", "0.2", "0.0"]]]
    doc = hostio.jplace_document(t, pl, " -p p -q x.fasta")
    js = json.loads(doc)  # the prettified text is still JSON
    assert list(js.keys()) == ["metadata", "tree", "placements", "fields", "version"]
    assert js["version"] == 3 and js["metadata"]["invocation"] == "viromeplacer -p p -q x.fasta"
    assert js["fields"] == ["edge_num", "likelihood", "like_weight_ratio", "distal_length", "pendant_length"]
    assert js["tree"] == t.jplace_newick()
    assert js["placements"][0] == {"p": [[0, -1.5, 0.75, 0.05, 0.0], [2, -2.25, 0.25, 0.15, 0.0]],
                                   "nm": [["r1 full header", 1], ["r1dup", 1]]}
    assert js["placements"][1]["nm"] == [["r/3", 1]]
    assert '"r\\/3"' in doc  # json-simple escapes the solidus
    # the line structure the reference's regex prettifier produces
    assert doc.startswith('{"metadata":{"invocation":"viromeplacer -p p -q x.fasta"},"tree":"((A:')
    assert ',\n"placements":\n[\n{\n\t"p":\n\t[[0,-1.5,0.75,0.05,0.0],\n\t[2,-2.25,0.25,0.15,0.0]],\n\t"nm":\n\t[[' in doc
    assert doc.endswith(']\n}\n],\n\n\t"fields":["edge_num","likelihood","like_weight_ratio","distal_length",'
                        '"pendant_length"],\n\t"version":3}')
    g = json.loads(hostio.jplace_document(t, hostio.jplace_placements(t, names, n_rows, branch, score, lwr, True), "", True))
    assert g["fields"] == ["distal_length", "edge_num", "like_weight_ratio", "likelihood", "pendant_length"]
    assert g["placements"][0]["p"][0] == [0.05, 0, 0.75, -1.5, 0.0]


def test_jsondb_roundtrip_and_tolerant_parse():
    db = synth.make_db(4, 4, 5, 40, 120, seed=3)
    nwk = "((A:0.1,B:0.2)C:0.3,D:0.4)R;"
    txt = hostio.dump_jsondb(db, nwk)
    assert '"states":core.DNAStatesShifted@' in txt  # as the reference writes it: not valid JSON
    with pytest.raises(json.JSONDecodeError):
        json.loads(txt)
    d = hostio.load_jsondb(txt)
    assert d["alphabet"] == 4 and d["k"] == 4 and d["n_branches"] == 5
    assert d["thr"] == db.thr and d["thr_log10"] == db.thr_log10
    assert np.array_equal(d["key_codes"], db.key_codes) and np.array_equal(d["row_offsets"], db.row_offsets)
    assert np.array_equal(d["branch_ids"], db.branch_ids)
    assert np.array_equal(d["scores"].view(np.uint32), db.scores.view(np.uint32))  # Float.toString round-trips bit-exactly
    assert d["tree"].jplace_newick() == hostio.parse_newick(nwk).jplace_newick()


def test_jsondb_rejects_non_dna_kmers():
    txt = ('{"k":2,"PPStarThreshold":0.1,"PPStarThresholdAsLog10":-1.0,"originalTree":"(A:1,B:1)R;",'
           '"hash":{"RH":{"1":-0.5}}}')
    with pytest.raises(ValueError, match="not a DNA"):
        hostio.load_jsondb(txt)


# ---- the --jsondb fixture authored from the Java (tests/golden/jsondb/make_jsondb_fixture.py), not from dump_jsondb ----
def _fixture_text():
    import os
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "jsondb")
    return open(os.path.join(here, "jsondb_toy.json")).read()


def test_jsondb_fixture_is_what_its_authoring_script_writes():
    import importlib.util
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "jsondb", "make_jsondb_fixture.py")
    spec = importlib.util.spec_from_file_location("make_jsondb_fixture", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.document() == _fixture_text()
    # java.lang.String.hashCode known answers the HashMap order rests on
    assert mod.java_string_hash("hello") == 99162322 and mod.java_string_hash("k") == 107


def test_jsondb_fixture_loads():
    """json-simple's layout as the reference writes it: HashMap key order, bare states / align tokens, Float.toString numbers
    (one in scientific notation), `null` for the infinite calibration score, `\\/` escapes"""
    txt = _fixture_text()
    assert txt.startswith('{"originalTree":"((A:0.1') and '"states":core.DNAStatesShifted@1b2c6ec2,' in txt
    d = hostio.load_jsondb(txt)
    assert d["alphabet"] == 4 and d["k"] == 3 and d["n_branches"] == 5 and d["calibration"] is None
    assert d["thr"] == np.float32(0.052734375) and d["thr_log10"] == np.float32(-1.2779074)
    rows = {}
    for r, code in enumerate(d["key_codes"].tolist()):
        a, b = int(d["row_offsets"][r]), int(d["row_offsets"][r + 1])
        rows[code] = list(zip(d["branch_ids"][a:b].tolist(), d["scores"][a:b].tolist()))
    code = lambda kmer: sum("ATCG".index(c) << (2 * i) for i, c in enumerate(kmer))
    f = lambda x: float(np.float32(x))
    assert rows == {
        code("ATC"): [(1, f(-0.30103)), (2, f(-0.04575749))],
        code("AAA"): [(2, -1.0), (4, f(-0.2218487))],
        code("GCA"): [(1, -0.75)],
        code("TTT"): [(1, f(-1.2779074)), (3, -0.0)],
        code("TCG"): [(2, -0.5), (3, -1.25), (4, -0.125)],
        code("GAT"): [(1, f(-1.2041199)), (3, f(-0.69897))],
        code("CGA"): [(4, -9.765625e-4)],
    }
    assert [int(c) for c in d["key_codes"]] == [code(k) for k in ("ATC", "AAA", "GCA", "TTT", "TCG", "GAT", "CGA")]  # dump order kept
    t = d["tree"]
    assert [n.label for n in t.nodes] == ["R", "C", "A", "B", "D"]
    # the engine accepts it as it is (argument checks + image construction, no device needed)
    from rappas_amd import placement
    info = placement.validate_db(4, 3, 5, d["thr_log10"], d["thr"], d["key_codes"], d["row_offsets"], d["branch_ids"], d["scores"])
    assert info.n_keys == 7 and info.n_entries == 13
