"""Developer check on a GPU box: engine vs oracle on small configs, all lane-group widths / table modes."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rappas_amd as ra
from rappas_amd import synth
from oracle import oracle as O
from tests.util import compare_with_oracle

def run(name, nreads, lanes=(0, 8, 16, 32, 64), modes=(ra.RK_TABLE_DIRECT, ra.RK_TABLE_DIRECT8, ra.RK_TABLE_HASH), scale=1.0, **rk):
    alphabet, k, leaves, _, _, rl, _ = synth.CONFIGS[name]
    sdb = synth.make_config_db(name, scale=scale)
    odb = O.OracleDB.from_synth(sdb)
    seq, off = synth.make_reads(alphabet, nreads, rl, **rk)
    ref = odb.place(seq, off)
    for mode in modes:
        db = ra.PhyloKmerDB.from_synth(sdb, table_mode=mode)
        pp = ra.PlacementProcess(db)
        for g in lanes:
            db.set_lanes_per_read(g)
            t = time.time()
            try:
                got = pp.processQueries(seq, off)
                st = compare_with_oracle(got, ref, odb, seq, off)
                print(f"{name} mode={mode} lanes={g}: OK {st} {time.time()-t:.2f}s  [{db.kernel_name()}]", flush=True)
            except AssertionError as e:
                print(f"{name} mode={mode} lanes={g}: MISMATCH\n{str(e)[:3000]}", flush=True)
        db.close()

if __name__ == "__main__":
    run("C1", 2000)
    run("C1", 2000, amb_rate=0.01, bad_rate=0.01, var_len=145)
    run("C4", 2000)
    run("C2", 5000, scale=0.1)
