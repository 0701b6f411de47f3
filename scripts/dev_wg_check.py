import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rappas_amd as ra
from rappas_amd import synth
from oracle import oracle as O
from tests.util import compare_with_oracle
case = sys.argv[1] if len(sys.argv) > 1 else "short"
if case == "short":
    sdb = synth.make_db(4, 7, 19999, 12000, 600_000, seed=8); nreads, rl = 20, 120
elif case == "long":
    sdb = synth.make_db(4, 6, 19999, 3000, 6_000_000, seed=11); nreads, rl = 20, 250
else:
    sdb = synth.make_db(4, 7, 30000, 8000, 2_000_000, seed=12); nreads, rl = 20, 200
seq, off = synth.make_reads(4, nreads, rl, seed=17)
odb = O.OracleDB.from_synth(sdb)
ref = odb.place(seq, off)
print("oracle done", ref["counters"], flush=True)
db = ra.PhyloKmerDB.from_synth(sdb)
print(db.kernel_name(), flush=True)
pp = ra.PlacementProcess(db)
t = time.time()
got = pp.processQueries(seq, off)
print("gpu done", time.time() - t, flush=True)
print(compare_with_oracle(got, ref, odb, seq, off), flush=True)
