"""(pinned buffers here come from torch.pin_memory(); rk_host_alloc gives a C caller the same)
PCIe-inclusive rate of the host-buffer entry point rk_place_batch (ASCII in, results out), pageable vs pinned memory."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rappas_amd as ra
from rappas_amd import synth, _lib
sdb = synth.make_config_db("C2")
db = ra.PhyloKmerDB.from_synth(sdb)
pp = ra.PlacementProcess(db)
n, K = 4_000_000, 7
seq, off = synth.make_reads(4, n, 150, seed=1)
pp.processQueries(seq[:150 * 100000], off[:100001])
for rep in range(3):  # the first full-size call also grows the engine's staging buffers
    t = time.perf_counter(); out = pp.processQueries(seq, off); dt = time.perf_counter() - t
    print(f"rk_place_batch, pageable buffers (call {rep}, fresh result arrays): {n/dt/1e6:.2f} Mreads/s ({dt*1e3:.0f} ms for {n} reads, {seq.nbytes/1e6:.0f} MB ASCII in, {n*99/1e6:.0f} MB out), placed={out.counters['placed']}")
# the same with result arrays that are reused (already mapped), as a JVM caller's would be
res_np = _lib.rk_result(*(x.ctypes.data_as(C.c_void_p) for x in (out.n_rows, out.branch, out.score, out.lwr, out.flags)))
p0 = _lib.rk_params(K, 0.01, 1, float("-inf")); ct0 = _lib.rk_counters()
lib0 = _lib.load()
for rep in range(2):
    t = time.perf_counter()
    _lib.check(lib0.rk_place_batch(db.handle, C.byref(p0), n, seq.ctypes.data_as(C.c_void_p), off.ctypes.data_as(C.c_void_p), C.byref(res_np), C.byref(ct0)))
    dt = time.perf_counter() - t
    print(f"rk_place_batch, pageable buffers, reused result arrays: {n/dt/1e6:.2f} Mreads/s ({dt*1e3:.0f} ms)")
# the same call on pinned (page-locked) caller buffers
lib = _lib.load()
pseq = torch.from_numpy(seq).pin_memory(); poff = torch.from_numpy(off.view(np.int64)).pin_memory()
o = dict(n_rows=torch.empty(n, dtype=torch.uint8).pin_memory(), branch=torch.empty((n, K), dtype=torch.int16).pin_memory(),
         score=torch.empty((n, K), dtype=torch.float32).pin_memory(), lwr=torch.empty((n, K), dtype=torch.float64).pin_memory(),
         flags=torch.empty(n, dtype=torch.int32).pin_memory())
res = _lib.rk_result(o["n_rows"].data_ptr(), o["branch"].data_ptr(), o["score"].data_ptr(), o["lwr"].data_ptr(), o["flags"].data_ptr())
p = _lib.rk_params(K, 0.01, 1, float("-inf")); ct = _lib.rk_counters()
for _ in range(2):
    t = time.perf_counter()
    _lib.check(lib.rk_place_batch(db.handle, C.byref(p), n, C.c_void_p(pseq.data_ptr()), C.c_void_p(poff.data_ptr()), C.byref(res), C.byref(ct)))
    dt = time.perf_counter() - t
print(f"rk_place_batch, pinned buffers:   {n/dt/1e6:.2f} Mreads/s ({dt*1e3:.0f} ms), placed={ct.placed}")
assert (o["score"].numpy().view(np.uint32) == out.score.view(np.uint32)).all()
