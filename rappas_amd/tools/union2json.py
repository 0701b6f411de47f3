"""`python -m rappas_amd.tools.union2json DB.union OUT.json`: a RAPPAS `.union` database (Java serialization) rewritten in the
layout of the reference's `--jsondb` dump (src/main_v2/SessionNext_v2.java:214-270), which the native driver `rk_place --jsondb`
reads.  DNA databases only: like the reference's own dump, the JSON layout spells k-mers as A/T/C/G strings
(AAStates.expandMer ignores its argument, so the reference cannot dump protein databases either); protein `.union` files go
through `python -m rappas_amd.tools.place --uniondb`."""
import sys

from .. import hostio
from ..synth import SynthDB


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    if len(argv) != 2:
        print(__doc__, file=sys.stderr)
        return 2
    with open(argv[0], "rb") as f:
        d = hostio.load_uniondb(f.read())
    if d["alphabet"] != 4:
        print("union2json: protein databases have no --jsondb form; use rappas_amd.tools.place --uniondb", file=sys.stderr)
        return 1
    db = SynthDB(4, d["k"], d["n_branches"], d["thr"], d["thr_log10"], d["key_codes"], d["row_offsets"], d["branch_ids"], d["scores"])
    with open(argv[1], "w") as f:
        f.write(hostio.dump_jsondb(db, hostio.write_newick(d["tree"], True, True, False), omega=d["omega"]))
    print(f"{len(d['key_codes'])} k-mers, {len(d['scores'])} entries -> {argv[1]}", file=sys.stderr)
    return 0


if __name__ == "__main__":
    sys.exit(main())
