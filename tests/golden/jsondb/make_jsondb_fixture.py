"""Authoring script of tests/golden/jsondb/jsondb_toy.json: a `--jsondb` dump written out BY HAND from the reference's Java, not by
rappas_amd.hostio.dump_jsondb (which the loaders were only ever tested against before).

What the reference does (src/main_v2/SessionNext_v2.java:214-270): a json-simple 1.1 `JSONObject` -- which `extends HashMap`
-- receives 16 `put`s and is streamed with `writeJSONString`.  Consequences reproduced here, each from the library's / JDK's
documented behaviour:

* key order = java.util.HashMap iteration order: bucket index `(h ^ (h >>> 16)) & (n - 1)` ascending, insertion order inside a
  bucket; `h = String.hashCode()` for the top level and the k-mer map, `Integer.hashCode() = value` for node-id maps; the table
  starts at 16 buckets and doubles whenever size exceeds 0.75 n (bins split in place, relative order kept);
* no white space: `{"key":value,"key":value}`;
* `states` and `align` hold objects whose classes do not override toString() and are not JSONAware, so json-simple emits
  `Object.toString()` bare: `core.DNAStatesShifted@<hex identity hash>` -- the file is NOT valid JSON there;
* Float values go through `Float.toString` unless infinite / NaN, which json-simple writes as `null`
  (`calibrationNormScore` is -Infinity for every database built with default options, Main_DBBUILD_3.java:1178);
* strings are escaped by JSONValue.escape (`/` becomes `\\/`);
* the Newick strings come from NewickWriter.getNewickTree(tree, true, true, false, withNodeIds) (src/tree/NewickWriter.java:116-212):
  branch lengths printed with 12 fraction digits; with node ids every label is prefixed `__id__` and -- a quirk of :191-195 --
  the id printed after a closing parenthesis is the LAST CHILD's, not the node's own;
* `hash` = Map<String, Map<Integer, Float>>: k-mer strings from DNAStatesShifted.expandMer, node ids as int, both HashMaps
  (Collectors.toMap).

The toy session: DNA, k = 3, omega = 1.5, tree ((A:0.1,B:0.2)C:0.3,D:0.4)R; (node ids in order of appearance R0 C1 A2 B3 D4,
NewickReader.java:76-160), seven k-mers.  Every float below is typed as Java prints it (shortest digits that identify the
float; scientific notation below 1e-3).  Run: python tests/golden/jsondb/make_jsondb_fixture.py
"""
import os

HERE = os.path.dirname(os.path.abspath(__file__))


def java_string_hash(s):
    h = 0
    for ch in s:
        h = (31 * h + ord(ch)) & 0xFFFFFFFF
    return h


def hashmap_order(keys, hash_fn):
    """iteration order of a java.util.HashMap after inserting `keys` (distinct) in the given order"""
    n = 16
    while len(keys) > 0.75 * n:
        n *= 2

    def bucket(k):
        h = hash_fn(k) & 0xFFFFFFFF
        return (h ^ (h >> 16)) & (n - 1)
    return sorted(keys, key=lambda k: (bucket(k), keys.index(k)))


def jobj(pairs, hash_fn):
    """{"k":v,...} in HashMap order; values are already JSON text"""
    d = dict(pairs)
    order = hashmap_order([k for k, _ in pairs], hash_fn)
    return "{" + ",".join('"%s":%s' % (k, d[k]) for k in order) + "}"


def jstr(s):
    return '"' + s.replace("\\", "\\\\").replace('"', '\\"').replace("/", "\\/") + '"'


# hash: k-mer -> {node id -> log10 PP*}; insertion order = the order fastutil's table would be walked in (arbitrary here)
HASH = [
    ("ATC", [(2, "-0.04575749"), (1, "-0.30103")]),
    ("TCG", [(2, "-0.5"), (3, "-1.25"), (4, "-0.125")]),
    ("CGA", [(4, "-9.765625E-4")]),
    ("GAT", [(1, "-1.2041199"), (3, "-0.69897")]),
    ("AAA", [(2, "-1.0"), (4, "-0.2218487")]),
    ("TTT", [(3, "-0.0"), (1, "-1.2779074")]),
    ("GCA", [(1, "-0.75")]),
]

TOP = [  # in the order of the put() calls, SessionNext_v2.java:220-262
    ("k", "3"),
    ("mink", "3"),
    ("omega", "1.5"),
    ("branchPerEdge", "1"),
    ("stateThreshold", "1.4E-45"),                 # Float.MIN_VALUE, SessionNext_v2.java:46
    ("PPStarThreshold", "0.052734375"),            # (float) Math.pow(1.5f / 4, 3), Main_DBBUILD_3.java:165
    ("PPStarThresholdAsLog10", "-1.2779074"),      # (float) Math.log10(0.052734375), :166
    ("states", "core.DNAStatesShifted@1b2c6ec2"),  # bare Object.toString()
    ("align", "alignement.Alignment@4edde6e5"),
    ("originalTree", jstr("((A:0.100000000000,B:0.200000000000)C:0.300000000000,D:0.400000000000)R;")),
    ("originalTreeWithNodeIds", jstr("((__2__A:0.100000000000,__3__B:0.200000000000)__3__C:0.300000000000,__4__D:0.400000000000)__4__R;")),
    ("extendedTree", jstr("((A:0.050000000000,B:0.100000000000)C:0.150000000000,D:0.200000000000)R;")),
    ("ARTree", jstr("((A:0.050000000000,B:0.100000000000)C/1:0.150000000000,D:0.200000000000)R;")),  # '/' -> '\\/'

    ("nodeMapping", jobj([(str(i), str(i)) for i in (4, 0, 3, 1, 2)], int)),
    ("calibrationNormScore", "null"),              # -Infinity -> json-simple writes null
    ("hash", jobj([(kmer, jobj([(str(n), v) for n, v in row], int)) for kmer, row in HASH], java_string_hash)),
]


def document():
    return jobj(TOP, java_string_hash)


if __name__ == "__main__":
    with open(os.path.join(HERE, "jsondb_toy.json"), "w") as f:
        f.write(document())
    print(document())
