"""Test-side emitter of the Java Object Serialization Stream Protocol (Java Object Serialization Specification, chapter 6),
used to assemble `.union`-shaped streams for rappas_amd/javaser.py and hostio.load_uniondb.

It writes what an ObjectOutputStream would write for objects DESCRIBED here (class descriptors with their field lists,
flags and super classes, default field data, writeObject annotations, back references); it has no connection to a JVM, and
the class layouts below follow the reference's sources for its own classes (field names / types of tree.PhyloNode,
tree.PhyloTree, core.hash.CustomHash_v4_FastUtil81 ...) and the documented serial forms of the JDK / fastutil containers.
The Swing super classes of PhyloTree are represented by stand-ins with the right structure (write-method annotations holding
block data and nested objects) -- enough to prove the reader walks through classes it knows nothing about.
"""
import struct

SC_WRITE_METHOD, SC_SERIALIZABLE = 0x01, 0x02
BASE = 0x7E0000


class CD:
    def __init__(self, name, uid, flags, fields, superdesc=None):
        self.name, self.uid, self.flags, self.fields, self.superdesc = name, uid, flags, fields, superdesc

    def chain(self):
        out, d = [], self
        while d:
            out.append(d)
            d = d.superdesc
        return out[::-1]


class Obj:
    def __init__(self, cd, values=None, annotations=None):
        self.cd, self.values, self.annotations = cd, values or {}, annotations or {}


class Arr:
    def __init__(self, cd, values):
        self.cd, self.values = cd, values


PRIM = {"B": ">b", "C": ">H", "D": ">d", "F": ">f", "I": ">i", "J": ">q", "S": ">h", "Z": ">?"}


class Writer:
    def __init__(self):
        self.out = bytearray(struct.pack(">HH", 0xACED, 5))
        self.handles = {}
        self.strings = {}
        self.keep = []  # every object that owns a handle stays alive: handles are keyed by id()
        self.n = 0

    def _new(self, key=None, owner=None):
        if key is not None:
            self.handles[key] = self.n
            self.keep.append(owner)
        self.n += 1

    def _utf(self, s):
        b = s.encode("utf-8")
        self.out += struct.pack(">H", len(b)) + b

    def block(self, data):
        """primitive data written by writeInt / writeFloat / ... in block-data mode"""
        data = bytes(data)
        while data:
            part, data = data[:255], data[255:]
            self.out += bytes([0x77, len(part)]) + part

    def block_long(self, data):
        self.out += bytes([0x7A]) + struct.pack(">i", len(data)) + bytes(data)

    def string(self, s):
        if s in self.strings:
            self.out += bytes([0x71]) + struct.pack(">i", BASE + self.strings[s])
            return
        self.out.append(0x74)
        self.strings[s] = self.n
        self._new()
        self._utf(s)

    def classdesc(self, cd):
        if cd is None:
            self.out.append(0x70)
            return
        if id(cd) in self.handles:
            self.out += bytes([0x71]) + struct.pack(">i", BASE + self.handles[id(cd)])
            return
        self.out.append(0x72)
        self._utf(cd.name)
        self.out += struct.pack(">q", cd.uid)
        self._new(id(cd), cd)
        self.out.append(cd.flags)
        self.out += struct.pack(">H", len(cd.fields))
        for f in cd.fields:
            self.out.append(ord(f[0]))
            self._utf(f[1])
            if f[0] in "[L":
                self.string(f[2])
        self.out.append(0x78)  # no class annotation
        self.classdesc(cd.superdesc)

    def obj(self, o):
        if o is None:
            self.out.append(0x70)
        elif isinstance(o, str):
            self.string(o)
        elif id(o) in self.handles:
            self.out += bytes([0x71]) + struct.pack(">i", BASE + self.handles[id(o)])
        elif isinstance(o, Arr):
            self.out.append(0x75)
            self.classdesc(o.cd)
            self._new(id(o), o)
            self.out += struct.pack(">i", len(o.values))
            t = o.cd.name[1]
            if t in PRIM:
                for v in o.values:
                    self.out += struct.pack(PRIM[t], v)
            else:
                for v in o.values:
                    self.obj(v)
        elif isinstance(o, Obj):
            self.out.append(0x73)
            self.classdesc(o.cd)
            self._new(id(o), o)
            for cd in o.cd.chain():
                vals = o.values.get(cd.name, {})
                for f in cd.fields:
                    if f[0] in PRIM:
                        self.out += struct.pack(PRIM[f[0]], vals[f[1]])
                    else:
                        self.obj(vals.get(f[1]))
                if cd.flags & SC_WRITE_METHOD:
                    for item in o.annotations.get(cd.name, []):
                        if isinstance(item, (bytes, bytearray)):
                            self.block(item)
                        else:
                            self.obj(item)
                    self.out.append(0x78)
        else:
            raise TypeError(type(o))


# ---- class descriptors (field order as ObjectStreamClass lists them: primitives by name, then object fields by name) ----
S, W = SC_SERIALIZABLE, SC_SERIALIZABLE | SC_WRITE_METHOD
NUMBER = CD("java.lang.Number", -8742448824652078965, S, [])
INTEGER = CD("java.lang.Integer", 1360826667806852920, S, [("I", "value")], NUMBER)
CHARACTER = CD("java.lang.Character", 3786198910865385080, S, [("C", "value")])
BYTE = CD("java.lang.Byte", -7183698231559129828, S, [("B", "value")], NUMBER)
HASHMAP = CD("java.util.HashMap", 362498820763181265, W, [("F", "loadFactor"), ("I", "threshold")])
VECTOR = CD("java.util.Vector", -2767605614048989439, W,
            [("I", "capacityIncrement"), ("I", "elementCount"), ("[", "elementData", "[Ljava/lang/Object;")])
OBJ_ARRAY = CD("[Ljava.lang.Object;", -8012369246846506644, S, [])
BYTE_ARRAY = CD("[B", -5984413125824719648, S, [])
CHAR_ARRAY = CD("[C", -5753798564021173076, S, [])

DMTN = CD("javax.swing.tree.DefaultMutableTreeNode", -4298474751201349152, W,
          [("Z", "allowsChildren"), ("L", "children", "Ljava/util/Vector;"), ("L", "parent", "Ljavax/swing/tree/MutableTreeNode;")])
PHYLONODE = CD("tree.PhyloNode", 2010, S,  # src/tree/PhyloNode.java:24-56
               [("F", "branchLengthToAncestor"), ("F", "branchLengthToOriginalAncestor"), ("F", "branchLengthToOriginalSon"),
                ("I", "id"), ("Z", "isFakeNode"), ("I", "jplaceEdgeId"), ("L", "label", "Ljava/lang/String;")], DMTN)
JCOMPONENT = CD("javax.swing.JComponent", -7908749299918704233, W, [("F", "alignmentX"), ("Z", "isAlignmentXSet"), ("L", "border", "Ljavax/swing/border/Border;")])
JTREE = CD("javax.swing.JTree", -1, W, [("Z", "rootVisible"), ("I", "rowHeight"), ("L", "treeModel", "Ljavax/swing/tree/TreeModel;")], JCOMPONENT)
TREEMODEL = CD("javax.swing.tree.DefaultTreeModel", -2, W, [("Z", "asksAllowsChildren"), ("L", "root", "Ljavax/swing/tree/TreeNode;")])
PHYLOTREE = CD("tree.PhyloTree", 2000, S,  # src/tree/PhyloTree.java:30-47
               [("Z", "isJplaceType"), ("Z", "isRooted"), ("I", "leavesCount"), ("I", "nodeCount"),
                ("L", "indexById", "Ljava/util/HashMap;"), ("L", "indexByName", "Ljava/util/HashMap;"),
                ("L", "orderedLeavesIds", "Ljava/util/ArrayList;")], JTREE)
ABSTRACTSTATES = CD("core.AbstractStates", 6000, S, [])
DNASTATES = CD("core.DNAStatesShifted", 6003, S,  # src/core/DNAStatesShifted.java:22-39
               [("I", "ambigousStatesCount"), ("L", "ambiguousState", "Ljava/util/HashMap;"), ("[", "bytes", "[B"),
                ("[", "maskArray", "[B"), ("[", "states", "[C")], ABSTRACTSTATES)
AASTATES = CD("core.AAStates", 6001, S, [("L", "ambiguousState", "Ljava/util/HashMap;"), ("L", "b", "Ljava/util/HashMap;"), ("L", "s", "Ljava/util/HashMap;")], ABSTRACTSTATES)
ALIGNMENT = CD("alignement.Alignment", 1000, S, [("I", "reducedAlignmentLength")])
HASHSTRATEGY = CD("core.hash.HashStrategy", 7001, S, [])
O2O_FUNC = CD("it.unimi.dsi.fastutil.objects.AbstractObject2ObjectFunction", -4940583368468432370, S, [("L", "defRetValue", "Ljava/lang/Object;")])
O2O_MAP = CD("it.unimi.dsi.fastutil.objects.AbstractObject2ObjectMap", -4940583368468432370, S, [], O2O_FUNC)
O2O = CD("it.unimi.dsi.fastutil.objects.Object2ObjectOpenCustomHashMap", 0, W,
         [("F", "f"), ("I", "size"), ("L", "strategy", "Lit/unimi/dsi/fastutil/Hash$Strategy;")], O2O_MAP)
C2F_FUNC = CD("it.unimi.dsi.fastutil.chars.AbstractChar2FloatFunction", -4940583368468432370, S, [("F", "defRetValue")])
C2F_MAP = CD("it.unimi.dsi.fastutil.chars.AbstractChar2FloatMap", -4940583368468432370, S, [], C2F_FUNC)
C2F = CD("it.unimi.dsi.fastutil.chars.Char2FloatOpenHashMap", 0, W, [("F", "f"), ("I", "size")], C2F_MAP)
CUSTOMHASH = CD("core.hash.CustomHash_v4_FastUtil81", 7000, S,  # src/core/hash/CustomHash_v4_FastUtil81.java:24-38
                [("I", "maxCapacitySize"), ("I", "nodeType"), ("L", "hash", "Lit/unimi/dsi/fastutil/objects/Object2ObjectOpenCustomHashMap;"),
                 ("L", "preparedNovelMap", "Lit/unimi/dsi/fastutil/chars/Char2FloatOpenHashMap;")])


def integer(v):
    return Obj(INTEGER, {"java.lang.Integer": {"value": v}})


def hashmap(pairs):
    items = [struct.pack(">ii", 16, len(pairs))]
    for k, v in pairs:
        items += [k, v]
    return Obj(HASHMAP, {"java.util.HashMap": {"loadFactor": 0.75, "threshold": 12}}, {"java.util.HashMap": items})


def vector(elems):
    arr = Arr(OBJ_ARRAY, list(elems) + [None] * 3)  # capacity beyond elementCount, as a real Vector carries
    return Obj(VECTOR, {"java.util.Vector": {"capacityIncrement": 0, "elementCount": len(elems), "elementData": arr}})


def phylo_tree(nodes_spec, rooted):
    """nodes_spec: list of (id, label, branch length, jplace edge id, parent id or None), children in list order"""
    objs = {}
    for i, label, bl, edge, parent in nodes_spec:
        objs[i] = Obj(PHYLONODE, {"tree.PhyloNode": dict(branchLengthToAncestor=bl, branchLengthToOriginalAncestor=-1.0, branchLengthToOriginalSon=-1.0,
                                                         id=i, isFakeNode=False, jplaceEdgeId=edge, label=label),
                                  DMTN.name: dict(allowsChildren=True, children=None, parent=None)},
                      {DMTN.name: [Arr(OBJ_ARRAY, ["userObject", label])]})
    for i, _, _, _, parent in nodes_spec:
        if parent is not None:
            objs[i].values[DMTN.name]["parent"] = objs[parent]
    for i in objs:
        kids = [objs[j] for j, _, _, _, p in nodes_spec if p == i]
        if kids:
            objs[i].values[DMTN.name]["children"] = vector(kids)
    root = next(objs[i] for i, _, _, _, p in nodes_spec if p is None)
    model = Obj(TREEMODEL, {TREEMODEL.name: dict(asksAllowsChildren=False, root=root)}, {TREEMODEL.name: [vector([])]})
    leaves = [i for i in objs if not any(p == i for _, _, _, _, p in nodes_spec)]
    return Obj(PHYLOTREE,
               {"tree.PhyloTree": dict(isJplaceType=False, isRooted=rooted, leavesCount=len(leaves), nodeCount=len(objs),
                                       indexById=hashmap([(integer(i), objs[i]) for i in sorted(objs)]), indexByName=None, orderedLeavesIds=None),
                JTREE.name: dict(rootVisible=True, rowHeight=16, treeModel=model),
                JCOMPONENT.name: dict(alignmentX=0.0, isAlignmentXSet=False, border=None)},
               {JTREE.name: [vector(["selectionModel", None])], JCOMPONENT.name: [struct.pack(">i", 0), None]})


def union_stream(alphabet, k, omega, thr, thr_log10, tree_obj, rows, convert_uo=False, calibration=float("-inf"), only_fakes=True,
                 long_blocks=False):
    """rows: list of (key bytes, [(node id, float), ...]); a row may also be its entries already packed big-endian (bytes of 6 per
    entry: u16 node id, f32 value), which is how the large streams of the memory test are written"""
    w = Writer()
    w.block(struct.pack(">iififff", k, k, omega, 1, 1.4e-45, thr, thr_log10))
    if alphabet == 4:
        states = Obj(DNASTATES, {DNASTATES.name: dict(ambigousStatesCount=11, ambiguousState=hashmap([]), bytes=Arr(BYTE_ARRAY, [0, 1, 2, 3]),
                                                      maskArray=Arr(BYTE_ARRAY, [3, 12, 48, -64]), states=Arr(CHAR_ARRAY, [65, 84, 67, 71]))})
    else:
        chars = [(Obj(CHARACTER, {CHARACTER.name: {"value": ord(c)}}), Obj(BYTE, {BYTE.name: {"value": i}})) for i, c in enumerate("RHKDESTNQCGPAILMFWYV")]
        if convert_uo:
            chars.append((Obj(CHARACTER, {CHARACTER.name: {"value": ord("U")}}), Obj(BYTE, {BYTE.name: {"value": 9}})))
        states = Obj(AASTATES, {AASTATES.name: dict(ambiguousState=hashmap([]), b=hashmap(chars), s=hashmap([]))})
    w.obj(states)
    w.obj(Obj(ALIGNMENT, {ALIGNMENT.name: {"reducedAlignmentLength": 60}}))
    w.obj(tree_obj)   # originalTree
    w.obj(tree_obj)   # extendedTree (a back reference here; a different tree in a real database)
    w.obj(tree_obj)   # ARTree
    w.obj(hashmap([(integer(i), integer(i)) for i in range(3)]))
    w.block(struct.pack(">f?", calibration, only_fakes))
    kv = []
    for key, row in rows:
        blob = bytes(row) if isinstance(row, (bytes, bytearray)) else b"".join(struct.pack(">Hf", n, v) for n, v in row)
        size = len(blob) // 6
        r = Obj(C2F, {C2F.name: {"f": 0.75, "size": size}, C2F_FUNC.name: {"defRetValue": 0.0}}, {C2F.name: [blob] if blob else []})
        kv += [Arr(BYTE_ARRAY, [b - 256 if b > 127 else b for b in key]), r]
    outer = Obj(O2O, {O2O.name: {"f": 0.8, "size": len(rows), "strategy": Obj(HASHSTRATEGY)}, O2O_FUNC.name: {"defRetValue": None}}, {O2O.name: kv})
    w.obj(Obj(CUSTOMHASH, {CUSTOMHASH.name: dict(maxCapacitySize=alphabet ** k, nodeType=2, hash=outer, preparedNovelMap=None)}))
    return bytes(w.out)
