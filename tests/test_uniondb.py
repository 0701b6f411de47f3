"""`.union` ingest without a JVM: rappas_amd/javaser.py (Java Object Serialization Stream Protocol reader)
and hostio.load_uniondb, on streams assembled from the specification by tests/javaser_writer.py.  PARITY UNPINNED: no file
written by a JVM exists here."""
import struct

import numpy as np
import pytest

from rappas_amd import hostio, javaser
from tests import javaser_writer as JW

# the toy session of tests/golden/jsondb/jsondb_toy.json: DNA k=3, tree ((A:0.1,B:0.2)C:0.3,D:0.4)R; ids R0 C1 A2 B3 D4,
# jplace edge ids post-order A0 B1 C2 D3 R4 (PhyloTree.java:408-439)
NODES = [(0, "R", 0.0, 4, None), (1, "C", 0.3, 2, 0), (2, "A", 0.1, 0, 1), (3, "B", 0.2, 1, 1), (4, "D", 0.4, 3, 0)]
CODE = lambda kmer: sum("ATCG".index(c) << (2 * i) for i, c in enumerate(kmer))
ROWS = [("ATC", [(1, -0.30103), (2, -0.04575749)]), ("TCG", [(2, -0.5), (3, -1.25), (4, -0.125)]), ("CGA", [(4, -9.765625e-4)]),
        ("AAA", []), ("TTT", [(1, -1.2779074), (3, -0.0)])]


def compress_mer(kmer):
    """DNAStatesShifted.compressMer (src/core/DNAStatesShifted.java:115-143): base i at bits 2*(i%4) of byte i/4"""
    out = bytearray((len(kmer) + 3) // 4)
    for i, c in enumerate(kmer):
        out[i // 4] |= "ATCG".index(c) << (2 * (i % 4))
    return bytes(out)


def toy_stream(**kw):
    rows = [(compress_mer(k), r) for k, r in ROWS if r]
    return JW.union_stream(4, 3, 1.5, 0.052734375, -1.2779074, JW.phylo_tree(NODES, True), rows, **kw)


def test_union_stream_loads_like_the_jsondb_of_the_same_session():
    d = hostio.load_uniondb(toy_stream())
    assert d["alphabet"] == 4 and d["k"] == 3 and d["n_branches"] == 5 and d["only_fakes"] is True
    assert d["thr"] == np.float32(0.052734375) and d["thr_log10"] == np.float32(-1.2779074) and d["calibration"] == float("-inf")
    want = [(k, r) for k, r in ROWS if r]
    assert d["key_codes"].tolist() == [CODE(k) for k, _ in want]
    assert d["row_offsets"].tolist() == [0, 2, 5, 6, 8]
    assert d["branch_ids"].tolist() == [n for _, r in want for n, _ in r]
    assert np.array_equal(d["scores"], np.array([v for _, r in want for _, v in r], np.float32))
    t = d["tree"]
    assert [n.label for n in t.nodes] == ["R", "C", "A", "B", "D"] and t.root.id == 0 and t.rooted
    assert [[c.id for c in n.children] for n in t.nodes] == [[1, 4], [2, 3], [], [], []]
    assert [n.jplace_edge for n in t.nodes] == [4, 2, 0, 1, 3]
    assert t.jplace_newick() == hostio.parse_newick("((A:0.1,B:0.2)C:0.3,D:0.4)R;").jplace_newick()
    # the same database read from the --jsondb fixture: same rows (the dump lists two more k-mers)
    import os
    j = hostio.load_jsondb(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "jsondb", "jsondb_toy.json")).read())
    jrows = {int(c): (j["branch_ids"][int(a):int(b)].tolist(), j["scores"][int(a):int(b)].tolist())
             for c, a, b in zip(j["key_codes"], j["row_offsets"][:-1], j["row_offsets"][1:])}
    for c, a, b in zip(d["key_codes"], d["row_offsets"][:-1], d["row_offsets"][1:]):
        assert jrows[int(c)] == (d["branch_ids"][int(a):int(b)].tolist(), d["scores"][int(a):int(b)].tolist())


def test_reader_walks_classes_it_does_not_know():
    recs = javaser.parse(toy_stream())
    kinds = [t for t, _ in recs]
    assert kinds == ["block", "object", "object", "object", "object", "object", "object", "block", "object"]
    tree = recs[3][1]
    assert tree.classname == "tree.PhyloTree" and recs[4][1] is tree and recs[5][1] is tree  # back references share one object
    assert [d.name for d in tree.desc.hierarchy()] == ["javax.swing.JComponent", "javax.swing.JTree", "tree.PhyloTree"]
    assert tree.get("nodeCount") == 5 and tree.get("rowHeight") == 16
    model = tree.get("treeModel")
    root = model.get("root")
    assert root.get("label") == "R" and root.get("parent") is None
    kids = javaser.arraylist_items(root.get("children"))
    assert [k.get("label") for k in kids] == ["C", "D"] and kids[0].get("parent") is root  # cycles resolved by handle
    assert tree.annotations["javax.swing.JComponent"] == [struct.pack(">i", 0), None]


def test_protein_database_and_convert_uo():
    nodes = [(0, "r", -1.0, 2, None), (1, "x", 0.5, 0, 0), (2, "y", 0.25, 1, 0)]
    rows = [(bytes([0, 19, 7]), [(1, -2.5)]), (bytes([9, 9, 14]), [(2, -0.25), (1, -3.0)])]
    for uo in (False, True):
        d = hostio.load_uniondb(JW.union_stream(20, 3, 1.5, 4.21875e-4, -3.3748, JW.phylo_tree(nodes, True), rows, convert_uo=uo))
        assert d["alphabet"] == 20 and d["convert_uo"] is uo
        assert d["key_codes"].tolist() == [0 | 19 << 5 | 7 << 10, 9 | 9 << 5 | 14 << 10]
        assert d["branch_ids"].tolist() == [1, 2, 1]


def test_long_rows_cross_block_data_records():
    """a row of 100 entries is 600 bytes of writeChar / writeFloat output: several TC_BLOCKDATA records of <= 255 bytes"""
    row = [(i + 1, -0.01 * i) for i in range(100)]
    nodes = [(0, "r", -1.0, 0, None)] + [(i, f"t{i}", 0.1, i, 0) for i in range(1, 102)]
    d = hostio.load_uniondb(JW.union_stream(4, 3, 1.5, 0.05, -1.3, JW.phylo_tree(nodes, False), [(compress_mer("GAT"), row)]))
    assert d["branch_ids"].tolist() == [n for n, _ in row] and d["n_branches"] == 102 and not d["tree"].rooted
    assert np.array_equal(d["scores"], np.array([v for _, v in row], np.float32))


def test_errors_carry_the_stream_offset():
    data = toy_stream()
    with pytest.raises(javaser.JavaSerializationError) as e:
        javaser.parse(data[:200])
    assert "truncated" in str(e.value) and e.value.offset <= 200
    with pytest.raises(javaser.JavaSerializationError):
        javaser.parse(b"\x00\x01\x02\x03")
    bad = bytearray(data)
    bad[4 + 2 + 28] = 0x7F  # the type code of the first object
    with pytest.raises(javaser.JavaSerializationError) as e:
        javaser.parse(bytes(bad))
    assert e.value.offset == 34 and "0x7f" in str(e.value)
    with pytest.raises(ValueError):
        hostio.load_uniondb(data[:4] + data[4:34])  # scalars only: no objects


def hostile_streams():
    """streams no JVM writes: a class descriptor naming itself as its super class, and arrays nested 5000 deep"""
    head = b"\xac\xed\x00\x05"
    desc = lambda name: b"\x72" + struct.pack(">H", len(name)) + name + b"\x00" * 8 + b"\x02\x00\x00\x78"  # Serializable, no fields
    own_super = head + b"\x73" + desc(b"A") + b"\x71\x00\x7e\x00\x00"
    loop_of_two = head + b"\x73" + desc(b"A") + desc(b"B") + b"\x71\x00\x7e\x00\x00"
    deep = head + b"\x75" + desc(b"[Ljava.lang.Object;") + b"\x70" + struct.pack(">i", 1) + (b"\x75\x71\x00\x7e\x00\x00" + struct.pack(">i", 1)) * 5000 + b"\x70"
    return {"own_super": own_super, "loop_of_two": loop_of_two, "deep": deep}


def test_hostile_streams_are_refused():
    h = hostile_streams()
    for name in ("own_super", "loop_of_two"):
        with pytest.raises(javaser.JavaSerializationError) as e:
            javaser.parse(h[name])
        assert "its own super class" in str(e.value), name
    with pytest.raises(javaser.JavaSerializationError) as e:
        javaser.parse(h["deep"])
    assert "nested too deep" in str(e.value)
    shallow = hostile_streams.__globals__["struct"].pack(">i", 1)
    ok = b"\xac\xed\x00\x05\x75\x72\x00\x13[Ljava.lang.Object;" + b"\x00" * 8 + b"\x02\x00\x00\x78\x70" + shallow + (b"\x75\x71\x00\x7e\x00\x00" + shallow) * 50 + b"\x70"
    (kind, arr), = javaser.parse(ok)
    depth = 0
    while arr is not None:
        arr, depth = arr[0], depth + 1
    assert kind == "object" and depth == 51


def test_union_database_is_accepted_by_the_engine():
    d = hostio.load_uniondb(toy_stream())
    from rappas_amd import placement
    info = placement.validate_db(d["alphabet"], d["k"], d["n_branches"], d["thr_log10"], d["thr"], d["key_codes"], d["row_offsets"],
                                 d["branch_ids"], d["scores"])
    assert info.n_keys == 4 and info.n_entries == 8


def test_union2json_round_trip(tmp_path):
    """.union -> --jsondb layout (what rk_place reads) -> the same database"""
    from rappas_amd.tools import union2json
    (tmp_path / "DB.union").write_bytes(toy_stream())
    assert union2json.main([str(tmp_path / "DB.union"), str(tmp_path / "db.json")]) == 0
    u = hostio.load_uniondb(toy_stream())
    j = hostio.load_jsondb((tmp_path / "db.json").read_text())
    for f in ("key_codes", "row_offsets", "branch_ids"):
        assert np.array_equal(u[f], j[f])
    assert np.array_equal(u["scores"].view(np.uint32), j["scores"].view(np.uint32))
    assert u["tree"].jplace_newick() == j["tree"].jplace_newick() and u["thr_log10"] == j["thr_log10"]
