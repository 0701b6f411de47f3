"""End to end through the host-side rows (FASTA -> dedup -> GPU placement -> jplace) against the oracle."""
import json

import numpy as np
import pytest

from oracle import oracle as O
from rappas_amd import hostio, synth
from rappas_amd.tools import place as place_tool

pytestmark = pytest.mark.gpu


def _fasta(db, n, length, seed):
    seq, off = synth.make_reads(db.alphabet, n, length, seed=seed, amb_rate=0.002)
    reads = [bytes(seq[int(off[i]):int(off[i + 1])]).decode() for i in range(n)]
    lines = []
    for i, r in enumerate(reads):
        lines.append(f">read{i} sample=x/{i}")
        lines += [r[j:j + 60] for j in range(0, len(r), 60)]
        if i % 10 == 3:  # a duplicate with a gap inserted and another header
            lines.append(f">dup{i} of read{i}")
            lines.append(r[:7] + "-" + r[7:])
    return "\n".join(lines) + "\n", reads


@pytest.mark.parametrize("n_nodes,amb", [(101, "mean"), (64, "max")])
def test_fasta_to_jplace_matches_oracle(tmp_path, n_nodes, amb):
    db = synth.make_db(4, 6, n_nodes, 3000, 20000, seed=11)
    nwk = synth.make_newick(n_nodes, seed=5)
    fasta, reads = _fasta(db, 200, 120, seed=9)
    (tmp_path / "db.json").write_text(hostio.dump_jsondb(db, nwk))
    (tmp_path / "q.fasta").write_text(fasta)
    out = tmp_path / "q.jplace"
    assert place_tool.main(["--jsondb", str(tmp_path / "db.json"), "--fasta", str(tmp_path / "q.fasta"), "--out", str(out),
                            "--amb", amb]) == 0
    js = json.loads(out.read_text())
    tree = hostio.parse_newick(nwk)
    assert js["tree"] == tree.jplace_newick() and js["version"] == 3

    # expectation from the oracle on the unique reads
    uniq, names = hostio.dedup_reads(hostio.read_fasta(fasta))
    assert [s for _, s in uniq] == reads  # the gapped duplicates collapse onto their originals
    seq, off = hostio.pack_batch(reads)
    odb = O.OracleDB.from_synth(db)
    ref = odb.place(seq, off, amb_mode=O.AMB_MEAN if amb == "mean" else O.AMB_MAX)
    want = [i for i in range(len(reads)) if ref["n_rows"][i]]
    assert len(js["placements"]) == len(want) > 150
    ties = 0
    for p, i in zip(js["placements"], want):
        assert p["nm"] == [[nm, 1] for nm in names[i]]
        n = int(ref["n_rows"][i])
        assert len(p["p"]) == n
        for j, row in enumerate(p["p"]):
            assert np.float32(row[1]) == ref["score"][i, j]  # Float.toString round-trips
            assert row[2] == pytest.approx(ref["lwr"][i, j], rel=1e-9) and row[4] == 0.0
            if ref["flags"][i] & O.RO_FLAG_TIE:
                ties += 1
                continue
            node = tree.nodes[int(ref["branch"][i, j])]
            assert row[0] == node.jplace_edge
            assert np.float32(row[3]) == node.bl / np.float32(2)
    assert names[3] == ["read3 sample=x/3", "dup3"]


def test_build_tool_then_place_tool(tmp_path):
    """posterior tables -> tools.build (GPU) -> --jsondb file -> tools.place (GPU) -> jplace: reads drawn from a node's most
    likely states come back on that node's branch."""
    from rappas_amd.tools import build as build_tool
    n_nodes, L, k = 21, 300, 8
    nwk = synth.make_newick(n_nodes, seed=2)
    states, pp, _ = synth.make_pp_tables(4, n_nodes, L, seed=31, peaked=0.97)
    nb = np.arange(n_nodes, dtype=np.uint16)
    np.savez(tmp_path / "pp.npz", states=states, pp_log10=pp, node_branch=nb)
    (tmp_path / "tree.nwk").write_text(nwk)
    assert build_tool.main(["--pp", str(tmp_path / "pp.npz"), "--tree", str(tmp_path / "tree.nwk"), "-k", str(k),
                            "--out", str(tmp_path / "db.json")]) == 0
    rng = np.random.default_rng(8)
    letters = np.frombuffer(b"ATCG", np.uint8)
    lines, truth = [], []
    for i in range(60):
        node = int(rng.integers(0, n_nodes))
        s = int(rng.integers(0, L - 120))
        lines += [f">q{i}", letters[states[node, s:s + 120, 0]].tobytes().decode()]
        truth.append(node)
    (tmp_path / "q.fasta").write_text("\n".join(lines) + "\n")
    assert place_tool.main(["--jsondb", str(tmp_path / "db.json"), "--fasta", str(tmp_path / "q.fasta"),
                            "--out", str(tmp_path / "q.jplace")]) == 0
    js = json.loads((tmp_path / "q.jplace").read_text())
    tree = hostio.parse_newick(nwk)
    edge_of = {n.id: n.jplace_edge for n in tree.nodes}
    by_name = {p["nm"][0][0]: p["p"][0][0] for p in js["placements"]}
    hits = sum(by_name.get(f"q{i}") == edge_of[truth[i]] for i in range(60))
    assert hits >= 57, hits


@pytest.mark.parametrize("extra", [[], ["--amb", "max", "--keep-at-most", "5"], ["--guppy-compat", "--keep-factor", "0.2"]])
def test_native_driver_writes_the_same_jplace(tmp_path, extra):
    """rk_place (C++, rappas_amd/csrc/host) and `python -m rappas_amd.tools.place` over the same files: byte-identical .jplace."""
    import subprocess
    from rappas_amd import build
    exe = build.build_host_tools()
    db = synth.make_db(4, 6, 75, 3500, 30000, seed=13)
    nwk = synth.make_newick(75, seed=6)
    fasta, _ = _fasta(db, 300, 140, seed=10)
    (tmp_path / "db.json").write_text(hostio.dump_jsondb(db, nwk))
    (tmp_path / "q.fasta").write_text(fasta)
    args = ["--jsondb", str(tmp_path / "db.json"), "--fasta", str(tmp_path / "q.fasta"), "--out", str(tmp_path / "out.jplace")] + extra
    assert place_tool.main(args) == 0
    py = (tmp_path / "out.jplace").read_bytes()
    (tmp_path / "out.jplace").unlink()
    r = subprocess.run([exe] + args, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    cpp = (tmp_path / "out.jplace").read_bytes()
    assert cpp == py and len(py) > 10000
    # rk_place's default is the all-threads host path (rk_fastio.hpp); the one-string-at-a-time path, MD5 digests for the dedup
    # (the reference's), other thread counts: the same file
    import re
    no_call = lambda b: re.sub(rb'"invocation":"[^"]*"', b'"invocation":""', b)  # (the jplace records its own command line)
    for more in (["--classic-io"], ["--md5-dedup", "--threads", "3"], ["--threads", "1"], ["--threads", "11", "--timing"]):
        (tmp_path / "out.jplace").unlink()
        r = subprocess.run([exe] + args + more, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        assert no_call((tmp_path / "out.jplace").read_bytes()) == no_call(py), more
    assert json.loads(r.stdout.splitlines()[-1])["reads"] == 330
    # ... and from the engine's own image file (written without a device by one tool, by the other from its handle): the only
    # difference is the invocation string the jplace records
    img = tmp_path / "db.rkimg"
    r = subprocess.run([exe, "--jsondb", str(tmp_path / "db.json"), "--save-dbimage", str(img)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    iargs = ["--dbimage", str(img), "--fasta", str(tmp_path / "q.fasta"), "--out", str(tmp_path / "img.jplace")] + extra
    assert place_tool.main(iargs) == 0
    py_img = (tmp_path / "img.jplace").read_bytes()
    (tmp_path / "img.jplace").unlink()
    r = subprocess.run([exe] + iargs, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert (tmp_path / "img.jplace").read_bytes() == py_img
    strip = lambda b: json.dumps({k: v for k, v in json.loads(b).items() if k != "metadata"}, sort_keys=True)
    assert strip(py_img) == strip(py)
    img2 = tmp_path / "db2.rkimg"
    assert place_tool.main(args + ["--save-dbimage", str(img2)]) == 0
    assert img2.read_bytes() == img.read_bytes()
    # both drivers also leave logs/notplaced_<query>.tsv (PlacementProcess.java:797-806), identical


def test_notplaced_log_is_written_by_both_drivers(tmp_path):
    import subprocess
    from rappas_amd import build
    exe = build.build_host_tools()
    db = synth.make_db(4, 6, 75, 300, 2000, seed=13)  # few keys: short reads can miss every row
    nwk = synth.make_newick(75, seed=6)
    (tmp_path / "db.json").write_text(hostio.dump_jsondb(db, nwk))
    seq, off = synth.make_reads(4, 400, 9, seed=3)
    reads = [bytes(seq[int(off[i]):int(off[i + 1])]).decode() for i in range(400)]
    fasta = "".join(f">r{i} note {i}\n{s}\n" for i, s in enumerate(reads))
    (tmp_path / "q.fasta").write_text(fasta)
    args = ["--jsondb", str(tmp_path / "db.json"), "--fasta", str(tmp_path / "q.fasta"), "--out", str(tmp_path / "out.jplace")]
    assert place_tool.main(args) == 0
    log = tmp_path / "logs" / "notplaced_q.fasta.tsv"
    py = log.read_text()
    odb = O.OracleDB.from_synth(db)
    ref = odb.place(seq, off)
    want = "".join(f"r{i} note {i}\n" for i in range(400) if not (ref["flags"][i] & 1))
    assert py == want and 0 < py.count("\n") < 400
    log.unlink()
    r = subprocess.run([exe] + args, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert log.read_text() == py


def test_place_tool_accepts_a_union_database(tmp_path):
    """--uniondb: the reference's own database format read without a JVM (rappas_amd/javaser.py), same .jplace as --jsondb"""
    from tests import javaser_writer as JW
    db = synth.make_db(4, 6, 75, 800, 6000, seed=21)
    nwk = synth.make_newick(75, seed=6)
    tree = hostio.parse_newick(nwk)
    spec = [(n.id, n.label, float(n.bl), n.jplace_edge, n.parent.id if n.parent is not None else None) for n in tree.nodes]
    # children must be listed in tree order: parse order == id order for this writer
    rows = []
    for r, code in enumerate(db.key_codes.tolist()):
        a, b = int(db.row_offsets[r]), int(db.row_offsets[r + 1])
        key = int(code).to_bytes(2, "little")  # compressMer of a 6-mer: 12 bits in 2 bytes
        rows.append((key, [(int(db.branch_ids[e]), float(db.scores[e])) for e in range(a, b)]))
    blob = JW.union_stream(4, 6, 1.5, float(db.thr), float(db.thr_log10), JW.phylo_tree(spec, tree.rooted), rows)
    (tmp_path / "DB.union").write_bytes(blob)
    (tmp_path / "db.json").write_text(hostio.dump_jsondb(db, nwk))
    fasta, _ = _fasta(db, 200, 120, seed=4)
    (tmp_path / "q.fasta").write_text(fasta)
    common = ["--fasta", str(tmp_path / "q.fasta")]
    assert place_tool.main(["--uniondb", str(tmp_path / "DB.union"), "--out", str(tmp_path / "u.jplace")] + common) == 0
    assert place_tool.main(["--jsondb", str(tmp_path / "db.json"), "--out", str(tmp_path / "j.jplace")] + common) == 0
    u = json.loads((tmp_path / "u.jplace").read_text())
    j = json.loads((tmp_path / "j.jplace").read_text())
    assert u["placements"] == j["placements"] and u["tree"] == j["tree"] and len(u["placements"]) > 100
    # the native driver reads the same file (rk_javaser.hpp): byte-identical .jplace
    import subprocess
    from rappas_amd import build
    args = ["--uniondb", str(tmp_path / "DB.union"), "--out", str(tmp_path / "u.jplace")] + common
    py = (tmp_path / "u.jplace").read_bytes()
    assert place_tool.main(args) == 0
    py = (tmp_path / "u.jplace").read_bytes()
    (tmp_path / "u.jplace").unlink()
    r = subprocess.run([build.build_host_tools()] + args, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert (tmp_path / "u.jplace").read_bytes() == py
