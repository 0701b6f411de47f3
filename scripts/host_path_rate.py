"""PCIe-inclusive rate of the host-buffer entry point rk_place_batch (ASCII in, results out, pageable memory)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rappas_amd as ra
from rappas_amd import synth
sdb = synth.make_config_db("C2")
db = ra.PhyloKmerDB.from_synth(sdb)
pp = ra.PlacementProcess(db)
n = 4_000_000
seq, off = synth.make_reads(4, n, 150, seed=1)
pp.processQueries(seq[:150 * 100000], off[:100001])
t = time.perf_counter(); out = pp.processQueries(seq, off); dt = time.perf_counter() - t
print(f"rk_place_batch host path: {n/dt/1e6:.2f} Mreads/s ({dt*1e3:.0f} ms for {n} reads, {seq.nbytes/1e6:.0f} MB ASCII in, {n*99/1e6:.0f} MB out), placed={out.counters['placed']}")
