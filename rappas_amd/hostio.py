"""Host-side callers / data formats either side of the placement hot path.

These are the pieces RAPPAS keeps in Java around the native call; they are restated here so that the engine can be
driven end-to-end without a JVM (`python -m rappas_amd.tools.place`).  No placement compute happens in this module.
Reference lines (paths relative to the reference root):

* N3  FASTA ingest   src/inputs/FASTAPointer.java:66-149 (multi-line records, blank and '#' lines skipped, no gap
                     stripping for placement: Main_PLACEMENT_v07.java:195), src/inputs/Fasta.java
      dedup          src/core/algos/PlacementProcess.java:591-629 (MD5 of the sequence without '-', case-sensitive; the first
                     occurrence keeps its FULL header, later duplicates are listed by the header cut at the first space)
* N2  --jsondb       src/main_v2/SessionNext_v2.java:214-270 (json-simple dump; `states` / `align` are bare toString() tokens,
                     so the file is not strictly JSON; only DNA dumps are usable: AAStates.expandMer ignores its argument)
* N1  jplace         src/main_v2/Main_PLACEMENT_v07.java:224-315, src/core/algos/PlacementProcess.java:1005-1046,
                     src/tree/NewickReader.java:46-160 (node ids in order of appearance = pre-order, root 0),
                     src/tree/PhyloTree.java:408-439 (jplace edge ids, post-order), src/tree/NewickWriter.java:116-212
"""
import hashlib
import json
import math
import re
import struct
from dataclasses import dataclass, field
from decimal import Decimal

import numpy as np

# ------------------------------------------------------------------------------------------------
# N3: FASTA + dedup
# ------------------------------------------------------------------------------------------------


def read_fasta(text):
    """-> list of (header, sequence).  `text`: str or bytes of a whole FASTA file.
    FASTAPointer.nextSequenceAsFasta: empty lines and lines starting with '#' are skipped, a line starting with '>'
    opens a record, sequence lines are concatenated as they are (gaps kept) and the result is trim()-med."""
    if isinstance(text, bytes):
        text = text.decode("utf-8", "replace")
    records, header, parts = [], None, []
    for line in text.splitlines():
        if not line or line.startswith("#"):
            continue
        if line[0] == ">":
            if header is not None:
                records.append((header, "".join(parts).strip()))
            header, parts = line[1:], []
        elif header is not None:
            parts.append(line)
    if header is not None:
        records.append((header, "".join(parts).strip()))
    return records


def dedup_reads(records):
    """PlacementProcess.java:591-629.  -> (unique [(header, sequence)], names [[str, ...]]) where names[i][0] is the full header
    of the first occurrence and names[i][1:] the space-cut headers of its duplicates, in file order."""
    index, unique, names = {}, [], []
    for header, seq in records:
        key = hashlib.md5(seq.replace("-", "").encode()).digest()
        if key in index:
            cut = header.find(" ")
            names[index[key]].append(header if cut < 0 else header[:cut])
        else:
            index[key] = len(unique)
            unique.append((header, seq))
            names.append([header])
    return unique, names


def notplaced_log(records, unique, placed):
    """Text of `notplaced_<query>.tsv` (Main_PLACEMENT_v07.java:214, PlacementProcess.java:797-806): the FULL header of every
    read none of whose k-mers is in the database, one per line, in file order.  The reference registers a checksum only for reads
    that produce a jplace record (:1046), so every later copy of an unplaced read is placed again, fails again and is logged
    again: all occurrences are listed, not just the first.  `placed[i]`: unique read i has RK_FLAG_PLACED."""
    index = {hashlib.md5(seq.replace("-", "").encode()).digest(): i for i, (_, seq) in enumerate(unique)}
    lines = [header for header, seq in records if not placed[index[hashlib.md5(seq.replace("-", "").encode()).digest()]]]
    return "".join(h + "\n" for h in lines)


def pack_batch(seqs):
    """list of str -> (uint8 concatenation, uint64 offsets) as rk_place_batch takes them."""
    off = np.zeros(len(seqs) + 1, np.uint64)
    off[1:] = np.cumsum([len(s) for s in seqs])
    buf = np.frombuffer("".join(seqs).encode("latin-1"), np.uint8) if off[-1] else np.zeros(0, np.uint8)
    return np.ascontiguousarray(buf), off


# ------------------------------------------------------------------------------------------------
# N1: tree (Newick) with the reference's node ids and jplace edge ids
# ------------------------------------------------------------------------------------------------
@dataclass
class Node:
    id: int
    label: str = ""
    bl: np.float32 = np.float32(0.0)
    children: list = field(default_factory=list)
    parent: "Node" = None
    jplace_edge: int = -1


@dataclass
class Tree:
    root: Node
    nodes: list  # by id

    @property
    def rooted(self):
        return len(self.root.children) == 2  # NewickReader.java:209-220

    def jplace_newick(self):
        return write_newick(self, True, True, True)


def tree_to_blob(tree):
    """the reference tree as the `user` blob of a database image (rk_db_save): exact, line based -- the twin of rkh::tree_to_blob
    (rappas_amd/csrc/host/rk_fastio.hpp):  RKTREE 1 <nodes> <root> / <id> <parent> <jplace edge> <float32 bits, hex> <n children> <ids...> TAB <label>"""
    out = [f"RKTREE 1 {len(tree.nodes)} {tree.root.id}\n"]
    for n in tree.nodes:
        bits = int(np.float32(n.bl).view(np.uint32))
        kids = "".join(f" {c.id}" for c in n.children)
        out.append(f"{n.id} {n.parent.id if n.parent is not None else -1} {n.jplace_edge} {bits:08x} {len(n.children)}{kids}\t{n.label}\n")
    return "".join(out).encode("utf-8")


def tree_from_blob(blob):
    text = blob.decode("utf-8")
    lines = text.split("\n")
    head = lines[0].split()
    if len(head) != 4 or head[0] != "RKTREE" or head[1] != "1":
        raise ValueError("database image carries no reference tree (RKTREE blob)")
    n_nodes, root = int(head[2]), int(head[3])
    nodes = [Node(i) for i in range(n_nodes)]
    for i in range(n_nodes):
        nums, label = lines[1 + i].split("\t", 1)
        f = nums.split()
        if int(f[0]) != i:
            raise ValueError("database image: malformed tree line")
        n = nodes[i]
        n.parent = nodes[int(f[1])] if int(f[1]) >= 0 else None
        n.jplace_edge = int(f[2])
        n.bl = np.array([int(f[3], 16)], dtype=np.uint32).view(np.float32)[0]
        n.children = [nodes[int(c)] for c in f[5:5 + int(f[4])]]
        n.label = label
    return Tree(nodes[root], nodes)


def parse_newick(s):
    """NewickReader.java:46-200: an internal node gets its id when its '(' is read, a leaf when its text ends; ids are
    therefore the pre-order numbering with the root = 0.  `label:length` is split at ':' (float32 length)."""
    s = s.strip()
    nodes, stack, cur_children = [], [], [[]]
    token, last_closed = [], None

    def fill(node, text):
        if not text:
            return
        data = text.split(":")
        node.label = data[0]
        if len(data) > 1 and data[1]:
            node.bl = np.float32(float(data[1].split("{")[0]))

    def new_node():
        n = Node(len(nodes))
        nodes.append(n)
        return n

    for ch in s:
        if ch == "(":
            stack.append(new_node())
            cur_children.append([])
            token, last_closed = [], None
        elif ch in ",);":
            text = "".join(token).strip()
            token = []
            if last_closed is not None:
                fill(last_closed, text)
                last_closed = None
            elif text or ch != ";":
                leaf = new_node()
                fill(leaf, text)
                cur_children[-1].append(leaf)
            if ch == ")":
                parent = stack.pop()
                for c in cur_children.pop():
                    c.parent = parent
                    parent.children.append(c)
                cur_children[-1].append(parent)
                last_closed = parent
            elif ch == ";":
                break
        else:
            token.append(ch)
    root = cur_children[0][0]
    tree = Tree(root, nodes)
    reset_jplace_edge_ids(tree)
    return tree


def reset_jplace_edge_ids(tree):
    """PhyloTree.java:408-439: depth-first; a leaf child gets the next id when met, an internal node after all of its
    children (the root last)."""
    counter = [-1]

    def dfs(node):
        for c in node.children:
            if not c.children:
                counter[0] += 1
                c.jplace_edge = counter[0]
            else:
                dfs(c)
        counter[0] += 1
        node.jplace_edge = counter[0]

    dfs(tree.root)


def _fmt12(x):
    """NumberFormat.getNumberInstance(Locale.UK) with exactly 12 fraction digits (NewickWriter.java:61-64): grouping commas,
    HALF_EVEN on the exact binary value."""
    return f"{Decimal(float(x)):,.12f}"


def write_newick(tree, with_bl, with_internal_names, with_jplace_labels):
    """NewickWriter.java:116-212 (no node-id prefix).  At an unrooted top level (3 sons) the root carries neither length
    nor edge label."""
    out = []

    def dfs(node, level):
        out.append("(")
        n = len(node.children)
        for i, c in enumerate(node.children):
            if not c.children:
                out.append(c.label)
                if with_bl:
                    out.append(":" + _fmt12(c.bl))
                if with_jplace_labels:
                    out.append("{%d}" % c.jplace_edge)
            else:
                dfs(c, level + 1)
            if i < n - 1:
                out.append(",")
            else:
                out.append(")")
                if with_internal_names:
                    out.append(node.label)
                if with_bl and level > -1:
                    out.append(":" + _fmt12(node.bl))
                if with_jplace_labels and level > -1:
                    out.append("{%d}" % node.jplace_edge)
        if node.parent is None:
            out.append(";")

    dfs(tree.root, 0 if tree.rooted else -1)
    return "".join(out)


# ------------------------------------------------------------------------------------------------
# N1: numbers the way json-simple prints them (Number.toString())
# ------------------------------------------------------------------------------------------------
def _java_float_repr(shortest, value):
    """Float.toString / Double.toString layout from the shortest uniquely-identifying decimal digits:
    plain decimal for 1e-3 <= |x| < 1e7, otherwise d.dddE[-]n; always at least one digit after the point.
    (JDK >= 19 digits.  Older JDKs print a few values with one digit more than the shortest form, e.g. 2.0E23 as
    1.9999999999999998E23 and Float.MIN_VALUE as 1.4E-45; both spellings parse to the same number.)"""
    if value != value or value in (float("inf"), float("-inf")):
        return "null"  # JSONValue.toJSONString: non-finite numbers become null
    d = Decimal(shortest)
    sign, digits, exp = d.as_tuple()
    digits = list(digits)
    while len(digits) > 1 and digits[-1] == 0:
        digits.pop()
        exp += 1
    neg = "-" if (sign or (value == 0 and math.copysign(1.0, value) < 0)) else ""
    if all(x == 0 for x in digits):
        return neg + "0.0"
    ds = "".join(map(str, digits))
    e10 = len(ds) + exp  # value = 0.ds * 10^e10
    a = abs(value)
    if 1e-3 <= a < 1e7:
        if e10 <= 0:
            body = "0." + "0" * (-e10) + ds
        elif e10 >= len(ds):
            body = ds + "0" * (e10 - len(ds)) + ".0"
        else:
            body = ds[:e10] + "." + ds[e10:]
    else:
        body = ds[0] + "." + (ds[1:] or "0") + "E" + str(e10 - 1)
    return neg + body


def java_double_to_string(x):
    return _java_float_repr(repr(float(x)), float(x))


def java_float_to_string(x):
    x = np.float32(x)
    return _java_float_repr(np.format_float_scientific(x, unique=True, trim="0") if np.isfinite(x) else "nan", float(x))


# ------------------------------------------------------------------------------------------------
# N1: jplace document
# ------------------------------------------------------------------------------------------------
def jplace_placements(tree, names, n_rows, branch, score, lwr, guppy=False):
    """One placement object per placed unique read, rows [edge_num, likelihood, like_weight_ratio, distal_length,
    pendant_length] (PlacementProcess.java:1005-1046); `names[i]` from dedup_reads."""
    out = []
    for i in range(len(n_rows)):
        if n_rows[i] == 0:
            continue
        rows = []
        for j in range(int(n_rows[i])):
            node = tree.nodes[int(branch[i, j])]
            edge = str(node.jplace_edge)
            like = java_float_to_string(score[i, j])
            ratio = java_double_to_string(lwr[i, j])
            distal = java_float_to_string(np.float32(node.bl) / np.float32(2))
            rows.append([distal, edge, ratio, like, "0.0"] if guppy else [edge, like, ratio, distal, "0.0"])
        out.append((rows, names[i]))
    return out


_SHORT_ESC = {'"': '\\"', "\\": "\\\\", "\b": "\\b", "\f": "\\f", "\n": "\\n", "\r": "\\r", "\t": "\\t", "/": "\\/"}


def _jstr(s):
    """JSONValue.escape of json-simple 1.1 (the reference's lib/json_simple-1.1.jar): the short escapes incl. '\\/', and
    \\uXXXX (upper-case hex) for U+0000-001F, U+007F-009F and U+2000-20FF; everything else verbatim."""
    out = ['"']
    for ch in s:
        if ch in _SHORT_ESC:
            out.append(_SHORT_ESC[ch])
        elif ch <= "\u001f" or "\u007f" <= ch <= "\u009f" or "\u2000" <= ch <= "\u20ff":
            out.append("\\u%04X" % ord(ch))
        else:
            out.append(ch)
    out.append('"')
    return "".join(out)


def jplace_document(tree, placements, call_string="", guppy=False):
    """Main_PLACEMENT_v07.java:224-315.  Key order follows json-simple's JSONObject (a java.util.HashMap): top level
    metadata, tree, placements, fields, version; a placement is p, nm.  The regex prettifier of :304-310 is applied."""
    fields = ["distal_length", "edge_num", "like_weight_ratio", "likelihood", "pendant_length"] if guppy else \
             ["edge_num", "likelihood", "like_weight_ratio", "distal_length", "pendant_length"]
    pl = []
    for rows, nm in placements:
        p = ",".join("[" + ",".join(r) + "]" for r in rows)
        n = ",".join("[" + _jstr(name) + ",1]" for name in nm)
        pl.append('{"p":[' + p + '],"nm":[' + n + "]}")
    out = ('{"metadata":{"invocation":' + _jstr("viromeplacer" + call_string) + '},"tree":' + _jstr(tree.jplace_newick()) +
           ',"placements":[' + ",".join(pl) + '],"fields":[' + ",".join(_jstr(f) for f in fields) + '],"version":3}')
    out = out.replace("},{", "\n},{\n\t")
    out = out.replace('],"', '],\n\t"')
    out = out.replace("]}],", "]\n}\n],\n")
    out = out.replace(',"placements":[{"p"', ',\n"placements":\n[\n{\n\t"p"')
    out = out.replace("],[", "],\n\t[")
    out = out.replace('"p":[[', '"p":\n\t[[')
    out = out.replace('"nm":[[', '"nm":\n\t[[')
    return out


# ------------------------------------------------------------------------------------------------
# N2: --jsondb dumps
# ------------------------------------------------------------------------------------------------
DNA_STATE = {"A": 0, "T": 1, "C": 2, "G": 3}


def load_jsondb(text):
    """SessionNext_v2.saveToJSON dump -> dict(alphabet, k, n_branches, thr, thr_log10, key_codes, row_offsets, branch_ids,
    scores, tree, calibration).  The bare `states` / `align` tokens are replaced by null before parsing."""
    if isinstance(text, bytes):
        text = text.decode("utf-8", "replace")
    text = re.sub(r'"(states|align)"\s*:\s*(?!["{\[\-\dntf])[^,}]+', r'"\1":null', text)
    doc = json.loads(text)
    k = int(doc["k"])
    tree = parse_newick(doc["originalTree"])
    codes, off, br, sc = [], [0], [], []
    for kmer, row in doc["hash"].items():
        if len(kmer) != k or any(c not in DNA_STATE for c in kmer):
            raise ValueError(f"jsondb: k-mer {kmer!r} is not a DNA {k}-mer (amino-acid dumps are unusable: "
                             "AAStates.expandMer ignores its argument)")
        codes.append(sum(DNA_STATE[c] << (2 * i) for i, c in enumerate(kmer)))
        for node, v in row.items():
            br.append(int(node))
            sc.append(v)
        off.append(len(br))
    return dict(alphabet=4, k=k, n_branches=len(tree.nodes), thr=np.float32(doc["PPStarThreshold"]),
                thr_log10=np.float32(doc["PPStarThresholdAsLog10"]), key_codes=np.array(codes, np.uint64),
                row_offsets=np.array(off, np.uint64), branch_ids=np.array(br, np.uint16), scores=np.array(sc, np.float32),
                tree=tree, calibration=doc.get("calibrationNormScore"), omega=doc.get("omega"))


def load_uniondb(data):
    """A `.union` database (SessionNext_v2.storeHash, src/main_v2/SessionNext_v2.java:109-147; read back by load, :158-207):
    a Java serialization stream holding, in this order, block data {int k, int minK, float omega, int branchPerEdge, float
    stateThreshold, float PPStarThreshold, float PPStarThresholdAsLog10}, the objects states, align, originalTree, extendedTree,
    ARTree, nodeMapping, block data {float calibrationNormScore, boolean onlyFakes} and the CustomHash_v4_FastUtil81.
    Same result dict as load_jsondb.  What is taken from the object graph:

    * alphabet: the class of `states` (core.DNAStatesShifted / core.AAStates; --convertUO shows as 'U' in AAStates' char map);
    * tree: PhyloTree.indexById (HashMap<Integer, PhyloNode>, src/tree/PhyloTree.java:39) -> per node id, label, branch length,
      jplace edge id (src/tree/PhyloNode.java:30-37) and, from its DefaultMutableTreeNode part, parent and ordered children;
    * rows: CustomHash_v4_FastUtil81.hash, an Object2ObjectOpenCustomHashMap<byte[], Char2FloatOpenHashMap>
      (src/core/hash/CustomHash_v4_FastUtil81.java:36): fastutil writes its open-hash maps as defaultWriteObject() followed by
      the entries -- writeObject(key), writeObject(value) for the outer map, writeChar(key), writeFloat(value) for a row -- `size`
      of them.  DNA keys are compressMer bytes (DNAStatesShifted.java:115-143: little-endian 2-bit codes), AA keys one state
      per byte.
    PARITY UNPINNED (rappas_amd/javaser.py): never run against a file written by a JVM."""
    from . import javaser
    recs = javaser.parse(data)
    blocks = b"".join(v for t, v in recs if t == "block")
    objs = [v for t, v in recs if t == "object"]
    if len(blocks) < 33 or len(objs) < 7:
        raise ValueError(f"union: expected 33 bytes of scalars and 7 objects, found {len(blocks)} and {len(objs)} (a database stored "
                         "without its hash?)")
    k, mink, omega, bpe, st_thr, thr, thr_log10 = struct.unpack(">iififff", blocks[:28])
    calib, only_fakes = struct.unpack(">f?", blocks[28:33])
    states, _align, otree, _etree, _artree, _nodemap, chash = objs[:7]
    if states.classname == "core.DNAStatesShifted":
        alphabet, convert_uo = 4, False
    elif states.classname == "core.AAStates":
        alphabet = 20
        b = states.get("b")
        convert_uo = b is not None and any(javaser.boxed(key) in (ord("U"), "U") for key, _ in javaser.hashmap_items(b))
    else:
        raise ValueError(f"union: unknown States class {states.classname}")
    # ---- original tree ----
    index = otree.get("indexById")
    if index is None:
        raise ValueError("union: originalTree has no indexById map")
    jn = {int(javaser.boxed(key)): val for key, val in javaser.hashmap_items(index)}
    nodes = [None] * len(jn)
    for i, o in jn.items():
        if not 0 <= i < len(nodes):
            raise ValueError(f"union: node id {i} outside 0..{len(nodes) - 1}")
        n = Node(i)
        n.label = o.get("label") or ""
        n.bl = np.float32(o.get("branchLengthToAncestor"))
        n.jplace_edge = int(o.get("jplaceEdgeId"))
        nodes[i] = n
    ident = {id(o): i for i, o in jn.items()}
    root = None
    for i, o in jn.items():
        ch = o.get("children")
        for c in (javaser.arraylist_items(ch) if ch is not None else []):
            nodes[i].children.append(nodes[ident[id(c)]])
            nodes[ident[id(c)]].parent = nodes[i]
    for n in nodes:
        if n.parent is None:
            root = n if root is None or n.id < root.id else root
    tree = Tree(root, nodes)
    # ---- hash ----
    outer = chash.get("hash")
    if outer is None:
        raise ValueError("union: CustomHash_v4_FastUtil81 without its map")
    oname = "it.unimi.dsi.fastutil.objects.Object2ObjectOpenCustomHashMap"
    rname = "it.unimi.dsi.fastutil.chars.Char2FloatOpenHashMap"
    kv = outer.objects(oname)
    n_keys = int(outer.get("size"))
    if len(kv) != 2 * n_keys:
        raise ValueError(f"union: outer map announces {n_keys} entries, stream holds {len(kv) // 2}")
    codes, off, br, sc = [], [0], [], []
    for key, row in zip(kv[0::2], kv[1::2]):
        raw = key.raw
        if alphabet == 4:
            code = int.from_bytes(raw, "little")  # compressMer bytes: base i at bits 2*(i%4) of byte i/4
            if code >> (2 * k):
                raise ValueError("union: DNA key has bits beyond 2k")
        else:
            code = sum((b & 0xFF) << (5 * i) for i, b in enumerate(raw))
        codes.append(code)
        m = int(row.get("size"))
        blob = row.block(rname)
        if len(blob) != 6 * m:
            raise ValueError(f"union: row announces {m} entries, stream holds {len(blob)} bytes")
        for e in range(m):
            node, v = struct.unpack_from(">Hf", blob, 6 * e)
            br.append(node)
            sc.append(v)
        off.append(len(br))
    return dict(alphabet=alphabet, convert_uo=convert_uo, k=k, n_branches=len(nodes), thr=np.float32(thr), thr_log10=np.float32(thr_log10),
                key_codes=np.array(codes, np.uint64), row_offsets=np.array(off, np.uint64), branch_ids=np.array(br, np.uint16),
                scores=np.array(sc, np.float32), tree=tree, calibration=calib, omega=omega, only_fakes=only_fakes)


def dump_jsondb(db, newick, omega=1.5):
    """Writer with the layout json-simple gives saveToJSON (for tests and for exchanging synthetic DBs): HashMap key order
    is not reproduced (irrelevant to any reader), floats are printed like Float.toString."""
    letters = "ATCG"
    rows = []
    for r in range(len(db.key_codes)):
        code = int(db.key_codes[r])
        kmer = "".join(letters[(code >> (2 * i)) & 3] for i in range(db.k))
        a, b = int(db.row_offsets[r]), int(db.row_offsets[r + 1])
        ent = ",".join(f'"{int(db.branch_ids[e])}":{java_float_to_string(db.scores[e])}' for e in range(a, b))
        rows.append(f'"{kmer}":{{{ent}}}')
    head = (f'"k":{db.k},"mink":{db.k},"omega":{java_float_to_string(omega)},"branchPerEdge":1,"stateThreshold":1.4E-45,'
            f'"PPStarThreshold":{java_float_to_string(db.thr)},"PPStarThresholdAsLog10":{java_float_to_string(db.thr_log10)},'
            f'"states":core.DNAStatesShifted@6d06d69c,"align":alignement.Alignment@7852e922,"originalTree":{_jstr(newick)},'
            f'"calibrationNormScore":null')
    return "{" + head + ',"hash":{' + ",".join(rows) + "}}"
