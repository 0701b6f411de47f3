"""`python -m rappas_amd.tools.place`: FASTA queries + a --jsondb dump -> .jplace, through the GPU engine.

The reference's `-p p` phase for one query file (src/main_v2/Main_PLACEMENT_v07.java:150-320) with the database taken
from the JSON dump `--jsondb` writes (src/main_v2/SessionNext_v2.java:214-270) instead of the Java-serialized .union.
Only the placement itself runs on the GPU; ingest and the jplace writer are rappas_amd/hostio.py.
"""
import argparse
import os
import sys

import numpy as np

from .. import hostio
from ..placement import PhyloKmerDB, PlacementProcess


def place_file(db_text, fasta_text, keep_at_most=7, keep_factor=0.01, amb="mean", ns_bound=float("-inf"), guppy=False,
               call_string="", device=0, union=False, dbimage=None, save_dbimage=None):
    """db_text: the bytes of a --jsondb dump, or (union=True) of a Java-serialized .union database; or dbimage = the path of the
    engine's own image file (rk_db_load: mmap + upload, the reference tree in its user blob)"""
    if dbimage is not None:
        from ..placement import db_image_info
        _, blob = db_image_info(dbimage)
        tree = hostio.tree_from_blob(blob)
        db = PhyloKmerDB.load(dbimage, device=device)
        if db.info.n_branches != len(tree.nodes):
            raise ValueError("database image: tree and database disagree on the number of branches")
    else:
        d = hostio.load_uniondb(db_text) if union else hostio.load_jsondb(db_text)
        tree = d["tree"]
        if save_dbimage is not None:
            from ..placement import save_db_image
            save_db_image(save_dbimage, d["alphabet"], d["k"], d["n_branches"], d["thr_log10"], d["thr"], d["key_codes"], d["row_offsets"],
                          d["branch_ids"], d["scores"], convert_uo=d.get("convert_uo", False), user=hostio.tree_to_blob(tree))
        db = PhyloKmerDB(d["alphabet"], d["k"], d["n_branches"], d["thr_log10"], d["thr"], d["key_codes"], d["row_offsets"],
                         d["branch_ids"], d["scores"], device=device, convert_uo=d.get("convert_uo", False))
    try:
        records = hostio.read_fasta(fasta_text)
        unique, names = hostio.dedup_reads(records)
        seq, off = hostio.pack_batch([s for _, s in unique])
        res = PlacementProcess(db, ns_bound).processQueries(
            seq, off, keepAtMost=keep_at_most, keepFactor=keep_factor, treatAmbiguities=(amb != "skip"),
            treatAmbiguitiesWithMax=(amb == "max"))
    finally:
        db.close()
    pl = hostio.jplace_placements(tree, names, res.n_rows, res.branch, res.score, res.lwr, guppy)
    res.notplaced = hostio.notplaced_log(records, unique, (res.flags & 1) != 0)
    return hostio.jplace_document(tree, pl, call_string, guppy), res


def main(argv=None):
    ap = argparse.ArgumentParser(prog="rappas_amd.tools.place", description=__doc__.splitlines()[0])
    g = ap.add_mutually_exclusive_group(required=True)
    g.add_argument("--jsondb", help="database dump written by the reference's --jsondb")
    g.add_argument("--uniondb", help="the reference's own database file (DB.union, Java serialization; SessionNext_v2.java:109-207)")
    g.add_argument("--dbimage", help="the engine's own database image (written by --save-dbimage / rk_db_save): mmap + upload, no parse")
    ap.add_argument("--save-dbimage", default=None, help="with --jsondb / --uniondb: also write the database as an image file")
    ap.add_argument("--fasta", required=True, help="query reads (-q)")
    ap.add_argument("--out", required=True, help="output .jplace")
    ap.add_argument("--keep-at-most", type=int, default=7)
    ap.add_argument("--keep-factor", type=float, default=0.01)
    ap.add_argument("--amb", choices=["mean", "max", "skip"], default="mean", help="--ambwithmax / --noamb")
    ap.add_argument("--nsbound", type=float, default=float("-inf"))
    ap.add_argument("--guppy-compat", action="store_true")
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--logs", default=None, help="directory of notplaced_<query>.tsv (default: logs/ next to --out, like the reference's workdir/logs)")
    a = ap.parse_args(argv)
    db_text = None
    if a.dbimage is None:
        with open(a.jsondb or a.uniondb, "rb") as f:
            db_text = f.read()
    with open(a.fasta, "rb") as f:
        fasta_text = f.read()
    call = "".join(" " + x for x in (argv if argv is not None else sys.argv[1:]))
    doc, res = place_file(db_text, fasta_text, a.keep_at_most, a.keep_factor, a.amb, a.nsbound, a.guppy_compat, call,
                          a.device, union=a.uniondb is not None, dbimage=a.dbimage, save_dbimage=a.save_dbimage)
    with open(a.out, "w") as f:
        f.write(doc)
    logs = a.logs if a.logs is not None else os.path.join(os.path.dirname(os.path.abspath(a.out)), "logs")
    os.makedirs(logs, exist_ok=True)
    with open(os.path.join(logs, "notplaced_" + os.path.basename(a.fasta) + ".tsv"), "w") as f:
        f.write(res.notplaced)
    placed = int(np.count_nonzero(res.n_rows))
    print(f"{len(res.n_rows)} unique reads, {placed} placed -> {a.out}", file=sys.stderr)
    return 0


if __name__ == "__main__":
    sys.exit(main())
