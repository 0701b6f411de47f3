"""The native host side (rappas_amd/csrc/host, `rk_place`) against its Python twin (rappas_amd/hostio.py): same trees,
numbers, dedup and database parse, byte for byte.  The pieces exercised here need no GPU."""
import hashlib
import subprocess

import numpy as np
import pytest

from rappas_amd import build, hostio, synth


@pytest.fixture(scope="module")
def rk_place():
    return build.build_host_tools()


def run(exe, *args):
    return subprocess.run([exe, *args], check=True, capture_output=True, text=True).stdout


@pytest.mark.parametrize("nwk", [
    "((A:0.1,B:0.2)C:0.3,D:0.4)R;",
    "(A:1,(B:2.5,C:1e-3):0.5,(D:1,E:1):1234.5);",
    "((a:0.000001,b:12345678.9)x:0.5,(c:3,d:4)y:0.25,e:7)root:0.0;",
    " ( ( t1:0.5 , t2:0.25 ) , t3 : 1 ) ; ",
])
def test_tree_numbering_and_jplace_string(rk_place, tmp_path, nwk):
    f = tmp_path / "t.nwk"
    f.write_text(nwk)
    out = run(rk_place, "--emit-tree", str(f)).splitlines()
    t = hostio.parse_newick(nwk)
    assert out[0] == t.jplace_newick()
    assert out[1] == hostio.write_newick(t, False, False, False)
    assert out[2] == ("rooted" if t.rooted else "unrooted")
    rows = [ln.split("\t") for ln in out[3:]]
    assert [(int(r[0]), r[1], int(r[2]), int(r[3])) for r in rows] == \
        [(n.id, n.label, n.jplace_edge, n.parent.id if n.parent is not None else -1) for n in t.nodes]


def test_random_tree_matches(rk_place, tmp_path):
    nwk = synth.make_newick(501, seed=4)
    f = tmp_path / "t.nwk"
    f.write_text(nwk)
    assert run(rk_place, "--emit-tree", str(f)).splitlines()[0] == hostio.parse_newick(nwk).jplace_newick()


def test_number_layouts(rk_place):
    rng = np.random.default_rng(1)
    floats = [0.0, -0.0, 1.0, 0.1, 1e-5, 3.4028235e38, 16777216.0, 0.05, -308.25, 0.33333334, 1e7, 9999999.0, 0.001, 9.999e-4]
    floats += list((rng.standard_normal(40) * 10.0 ** rng.integers(-12, 12, 40)).astype(np.float32))
    for x in floats:
        x = np.float32(x)
        assert run(rk_place, "--format-float", repr(float(x))).strip() == hostio.java_float_to_string(x), x
    doubles = [0.0, 0.5, 1e-300, 0.1 + 0.2, 123.456, 1e7, 1.2345e10, 1e-3, 1e-4, float("nan"), float("inf")]
    doubles += list(rng.standard_normal(40) * 10.0 ** rng.integers(-30, 30, 40))
    for x in doubles:
        assert run(rk_place, "--format-double", repr(float(x))).strip() == hostio.java_double_to_string(x), x


def test_md5_and_dedup(rk_place, tmp_path):
    for s in ["", "a", "ACGT" * 100, "x" * 55, "y" * 56, "w" * 57, "v" * 63, "z" * 64, "u" * 65, "t" * 119, "s" * 120, "r" * 128, "q" * 150]:
        assert run(rk_place, "--md5", s).strip() == hashlib.md5(s.encode()).hexdigest()
    txt = "# c\n>r1 desc\nACGT\n\nAC-GT\n>r2\n  ACGTN  \n>r3 x y\nACGTACGT\n>empty\n>r4\nacgt\n>r5 dup of r1\nACGTACG-T\r\n"
    f = tmp_path / "q.fa"
    f.write_text(txt)
    uniq, names = hostio.dedup_reads(hostio.read_fasta(txt))
    got = [ln.split("\t") for ln in run(rk_place, "--dedup", str(f)).split("\n")[:-1]]
    assert got == [[s] + n for (_, s), n in zip(uniq, names)]
    assert names[0] == ["r1 desc", "r3", "r5"]


def test_jsondb_parse(rk_place, tmp_path):
    db = synth.make_db(4, 5, 9, 150, 700, seed=5)
    nwk = synth.make_newick(9, seed=1)
    f = tmp_path / "db.json"
    f.write_text(hostio.dump_jsondb(db, nwk))
    out = run(rk_place, "--load-jsondb", str(f)).splitlines()
    d = hostio.load_jsondb(f.read_text())
    head = out[0].split()
    assert int(head[0]) == d["k"] and head[1] == hostio.java_float_to_string(d["thr"]) and head[2] == hostio.java_float_to_string(d["thr_log10"])
    assert int(head[3]) == len(d["key_codes"]) and int(head[4]) == len(d["scores"])
    assert out[1] == nwk
    for r, line in enumerate(out[2:]):
        parts = line.split()
        a, b = int(d["row_offsets"][r]), int(d["row_offsets"][r + 1])
        assert int(parts[0]) == int(d["key_codes"][r])
        assert parts[1:] == [f"{int(d['branch_ids'][e])}:{hostio.java_float_to_string(d['scores'][e])}" for e in range(a, b)]


def test_string_escaping_follows_json_simple():
    assert hostio._jstr('a/b"c\\d\n\x01\x7f é') == '"a\\/b\\"c\\\\d\\n\\u0001\\u007F\\u2028é"'


def test_native_loader_reads_the_fixture_authored_from_the_java(rk_place):
    """tests/golden/jsondb/jsondb_toy.json (written by hand from SessionNext_v2.saveToJSON + json-simple, not by dump_jsondb):
    the C++ loader gives the rows typed into the authoring script, in the dump's order"""
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "jsondb", "jsondb_toy.json")
    out = run(rk_place, "--load-jsondb", path).splitlines()
    assert out[0] == "3 0.052734375 -1.2779074 7 13"
    assert out[1] == "((A:0.100000000000,B:0.200000000000)C:0.300000000000,D:0.400000000000)R;"
    code = lambda kmer: sum("ATCG".index(c) << (2 * i) for i, c in enumerate(kmer))
    assert out[2:] == [
        f"{code('ATC')} 1:-0.30103 2:-0.04575749",
        f"{code('AAA')} 2:-1.0 4:-0.2218487",
        f"{code('GCA')} 1:-0.75",
        f"{code('TTT')} 1:-1.2779074 3:-0.0",
        f"{code('TCG')} 2:-0.5 3:-1.25 4:-0.125",
        f"{code('GAT')} 1:-1.2041199 3:-0.69897",
        f"{code('CGA')} 4:-9.765625E-4",
    ]


def _union_dump(d):
    """what `rk_place --load-uniondb` prints, from the Python reader's result"""
    t = d["tree"]
    f = hostio.java_float_to_string
    head = (f"{d['alphabet']} {d['k']} {int(d['convert_uo'])} {int(d['only_fakes'])} {f(d['thr'])} {f(d['thr_log10'])} {f(np.float32(d['omega']))} "
            f"{f(np.float32(d['calibration']))} {len(d['key_codes'])} {len(d['scores'])}")
    lines = [head, t.jplace_newick()]
    lines += [f"{n.id}\t{n.label}\t{n.jplace_edge}\t{n.parent.id if n.parent is not None else -1}" for n in t.nodes]
    for r in range(len(d["key_codes"])):
        a, b = int(d["row_offsets"][r]), int(d["row_offsets"][r + 1])
        ent = "".join(f" {int(d['branch_ids'][e])}:{f(d['scores'][e])}" for e in range(a, b))
        lines.append(f"{int(d['key_codes'][r])}{ent}")
    return "\n".join(lines) + "\n"


def test_union_reader_matches_its_python_twin(rk_place, tmp_path):
    """rk_javaser.hpp + load_uniondb (C++) against rappas_amd/javaser.py + hostio.load_uniondb on streams assembled from the
    serialization specification (tests/javaser_writer.py): DNA and protein sessions, a larger random tree, back references,
    the Swing classes the tree drags in.  Both readers are unpinned (no JVM here): this only shows they agree."""
    from tests import javaser_writer as JW
    from tests import test_uniondb as TU
    streams = {"toy": TU.toy_stream()}
    nodes = [(0, "r", -1.0, 2, None), (1, "x", 0.5, 0, 0), (2, "y", 0.25, 1, 0)]
    rows = [(bytes([0, 19, 7]), [(1, -2.5)]), (bytes([9, 9, 14]), [(2, -0.25), (1, -3.0)])]
    for uo in (False, True):
        streams[f"aa{int(uo)}"] = JW.union_stream(20, 3, 1.5, 4.21875e-4, -3.3748, JW.phylo_tree(nodes, True), rows, convert_uo=uo)
    db = synth.make_db(4, 6, 75, 800, 6000, seed=21)
    tree = hostio.parse_newick(synth.make_newick(75, seed=6))
    spec = [(n.id, n.label, float(n.bl), n.jplace_edge, n.parent.id if n.parent is not None else None) for n in tree.nodes]
    big = []
    for r, code in enumerate(db.key_codes.tolist()):
        a, b = int(db.row_offsets[r]), int(db.row_offsets[r + 1])
        big.append((int(code).to_bytes(2, "little"), [(int(db.branch_ids[e]), float(db.scores[e])) for e in range(a, b)]))
    streams["big"] = JW.union_stream(4, 6, 1.5, float(db.thr), float(db.thr_log10), JW.phylo_tree(spec, tree.rooted), big)
    for name, blob in streams.items():
        f = tmp_path / f"{name}.union"
        f.write_bytes(blob)
        assert run(rk_place, "--load-uniondb", str(f)) == _union_dump(hostio.load_uniondb(blob)), name


def test_union_reader_rejects_what_is_not_a_union(rk_place, tmp_path):
    from tests import test_uniondb as TU
    good = TU.toy_stream()
    hostile = TU.hostile_streams()  # a descriptor that is its own super class (two shapes), arrays nested 5000 deep
    for name, blob in (("magic", b"\x00\x01\x02\x03" + good[4:]), ("cut", good[:len(good) // 2]), ("empty", b""), *hostile.items()):
        f = tmp_path / f"{name}.union"
        f.write_bytes(blob)
        r = subprocess.run([rk_place, "--load-uniondb", str(f)], capture_output=True, text=True)
        assert r.returncode == 1 and "rk_place:" in r.stderr, (name, r.stderr)
        if name in ("own_super", "loop_of_two"):
            assert "its own super class" in r.stderr
        if name == "deep":
            assert "nested more than" in r.stderr
        with pytest.raises(ValueError):
            hostio.load_uniondb(blob)


# ---- round 4: the all-threads host path of rk_place (rk_fastio.hpp) is held to the one-string-at-a-time path above ----
def _messy_fasta(rng, n, dup_every=5, multiline=True, weird_names=False):
    """records with duplicates (incl. gap variants: the checksum strips '-'), blank / '#' / CRLF lines, multi-line sequences, leading
    and trailing blanks, headers with spaces, an empty sequence, text before the first record"""
    seqs = ["".join("ACGT"[int(b)] for b in rng.integers(0, 4, int(rng.integers(20, 220)))) for _ in range(n)]
    lines = ["stray line before any record", "", "# a comment"]
    for i in range(n):
        s = seqs[i]
        if i % dup_every == 3:
            s = seqs[int(rng.integers(0, i))]                      # a duplicate of an earlier read
            if i % 2:
                s = s[:7] + "-" + s[7:] + "--"                       # ... written with gaps
        name = f"read{i} sample={i % 3} len={len(s)}"
        if weird_names and i % 11 == 0:
            name = f"odd]name}},{{{i}],\"x\" [a],[b] /slash \\ back"
        lines.append(">" + name + ("\r" if i % 7 == 0 else ""))
        if i == 17:
            continue                                                # header only: an empty sequence
        if multiline and i % 3 == 0:
            w = int(rng.integers(10, 60))
            parts = [s[j:j + w] for j in range(0, len(s), w)]
            for j, part in enumerate(parts):
                lines.append(("  " if j == 0 and i % 6 == 0 else "") + part + ("\r" if i % 7 == 0 else ""))
                if j == 0 and i % 9 == 0:
                    lines += ["", "# inside a record"]
        else:
            lines.append(s + (" \t" if i % 4 == 0 else ""))
    return "\n".join(lines) + ("\n" if n % 2 else "")


@pytest.mark.parametrize("threads", [1, 3, 8])
@pytest.mark.parametrize("md5", [False, True])
def test_fast_scan_and_dedup_equal_the_classic_path(rk_place, tmp_path, threads, md5):
    rng = np.random.default_rng(threads)
    f = tmp_path / "q.fa"
    f.write_text(_messy_fasta(rng, 700))
    classic = run(rk_place, "--dedup", str(f))
    fast = run(rk_place, "--threads", str(threads), *(["--md5-dedup"] if md5 else []), "--dedup-fast", str(f))
    assert fast == classic and classic.count("\n") < 700 and classic.count("\t") == 700
    # one long record, no trailing newline, a file of comments only
    (tmp_path / "one.fa").write_text(">only\nACGT\nAC-GT")
    assert run(rk_place, "--threads", str(threads), "--dedup-fast", str(tmp_path / "one.fa")) == run(rk_place, "--dedup", str(tmp_path / "one.fa")) == "ACGTAC-GT\tonly\n"
    (tmp_path / "none.fa").write_text("# nothing\n\n")
    assert run(rk_place, "--threads", str(threads), "--dedup-fast", str(tmp_path / "none.fa")) == ""


@pytest.mark.parametrize("threads,guppy,K,weird", [(1, False, 7, False), (4, False, 7, False), (7, True, 3, False), (5, False, 16, True), (2, True, 1, True)])
def test_fast_jplace_writer_equals_the_classic_document(rk_place, tmp_path, threads, guppy, K, weird):
    """the text the reference's seven regex replacements leave (Main_PLACEMENT_v07.java:301-315), emitted directly by every thread for
    a range of reads, against the whole-document path; headers that the replacements could reach into take the exact path"""
    rng = np.random.default_rng(K + threads)
    f, t = tmp_path / "q.fa", tmp_path / "t.nwk"
    f.write_text(_messy_fasta(rng, 900, weird_names=weird))
    t.write_text(synth.make_newick(120, seed=K))
    a, b = tmp_path / "fast.jplace", tmp_path / "classic.jplace"
    for seed in (1, 2):
        out = run(rk_place, "--threads", str(threads), "--keep-at-most", str(K), *(["--guppy-compat"] if guppy else []),
                  "--write-selftest", str(f), str(t), str(a), str(b), str(seed)).split()
        assert a.read_bytes() == b.read_bytes()
        assert out[2] == ("exact" if weird else "direct") and int(out[1]) > 150
    # nothing placed at all
    (tmp_path / "e.fa").write_text("# empty\n")
    run(rk_place, "--write-selftest", str(tmp_path / "e.fa"), str(t), str(a), str(b), "3")
    assert a.read_bytes() == b.read_bytes()


def test_database_image_from_the_tool_without_a_gpu(rk_place, tmp_path):
    """rk_place --jsondb ... --save-dbimage: the image is built on the host (rk_db_save_desc) and carries the reference tree"""
    import rappas_amd as ra
    db = synth.make_db(4, 6, 21, 600, 4000, seed=8)
    nwk = synth.make_newick(21, seed=2)
    tree = hostio.parse_newick(nwk)
    assert len(tree.nodes) == 21
    (tmp_path / "db.json").write_text(hostio.dump_jsondb(db, nwk))
    img = tmp_path / "db.rkimg"
    subprocess.run([rk_place, "--jsondb", str(tmp_path / "db.json"), "--save-dbimage", str(img)], check=True, capture_output=True)
    info, blob = ra.db_image_info(str(img))
    assert (info.k, info.n_branches, info.n_keys, info.n_entries) == (6, 21, db.n_keys, db.n_entries)
    assert blob.startswith(b"RKTREE 1 21 0\n") and blob.count(b"\n") == 22


def test_union_reader_streams_a_ten_million_entry_database(rk_place, tmp_path):
    """a `.union` stream of 10^7 row entries in 3 x 10^5 rows (tests/javaser_writer.py; alignment, extended tree, AR tree and node
    mapping in front of the hash as SessionNext_v2.storeHash writes them): the native reader turns the rows into CSR as it reads them
    and lets go of the objects a placement has no use for -- peak resident memory stays a small multiple of the file (the round-3
    reader held a node graph per row: gigabytes here, tens of GB for a 10^7-ROW session)"""
    from tests import javaser_writer as JW
    n_rows, per_row, k = 300_000, 33, 10
    rng = np.random.default_rng(3)
    codes = rng.choice(4 ** k, size=n_rows, replace=False).astype(np.uint64)
    ent = np.zeros(n_rows * per_row, dtype=np.dtype([("b", ">u2"), ("v", ">f4")]))
    ent["b"] = (np.arange(n_rows * per_row) % per_row) + 1 + np.repeat(rng.integers(0, 900, n_rows), per_row)
    ent["v"] = -rng.random(n_rows * per_row, dtype=np.float32) * 4
    raw = ent.tobytes()
    tree = hostio.parse_newick(synth.make_newick(999, seed=6))
    spec = [(n.id, n.label, float(n.bl), n.jplace_edge, n.parent.id if n.parent is not None else None) for n in tree.nodes]
    rows = [(int(c).to_bytes(3, "little"), raw[i * per_row * 6:(i + 1) * per_row * 6]) for i, c in enumerate(codes.tolist())]
    blob = JW.union_stream(4, k, 1.5, 1e-5, -5.0, JW.phylo_tree(spec, tree.rooted), rows)
    path = tmp_path / "big.union"
    path.write_bytes(blob)
    size = len(blob)
    del rows, blob, raw, ent
    # (ru_maxrss survives exec: a process starts with its parent's resident size as its "peak".  The tool is therefore started by a
    #  small interpreter of its own, not by this one, which holds the stream it has just written)
    import sys
    out = run(sys.executable, "-c", "import subprocess, sys; sys.exit(subprocess.run(sys.argv[1:]).returncode)", rk_place, "--uniondb-stats", str(path)).split()
    assert (int(out[0]), int(out[1]), int(out[2])) == (n_rows, n_rows * per_row, 999)
    rss = int(out[3]) * 1024
    # the file once (read whole), its CSR form (8 + 8 bytes a row, 6 an entry; vectors grow by doubling) and little else
    assert size > 60e6 and rss < 5 * size and rss < 400e6, (size, rss)
