"""Developer measurement: PCIe-inclusive rates of the host entry points on C2 (pageable vs page-locked buffers, ASCII vs packed),
with RK_HOST_TIMING=1 the library prints where its host thread waits."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import rappas_amd as ra
from rappas_amd import synth, _lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
sdb = synth.make_config_db("C2")
db = ra.PhyloKmerDB.from_synth(sdb)
pp = ra.PlacementProcess(db)
seq, off = synth.make_reads(4, n, 150, seed=1)
K = 7
reuse = ra.Placements(np.zeros(n, np.uint8), np.zeros((n, K), np.uint16), np.zeros((n, K), np.float32), np.zeros((n, K), np.float64), np.zeros(n, np.uint32), {})
t0 = time.perf_counter(); packed, lens, flags = pp.pack_reads_host(seq, off); t1 = time.perf_counter()
print(f"rk_pack_reads_host (incl. allocating its outputs): {n / (t1 - t0) / 1e6:.1f} Mreads/s")
lib = _lib.load()
pk2 = np.zeros_like(packed); l2 = np.zeros_like(lens); f2 = np.zeros_like(flags)
for th in (1, 4, 8, 16):
    t0 = time.perf_counter()
    lib.rk_pack_reads_host(db.handle, n, seq.ctypes.data_as(C.c_void_p), off.ctypes.data_as(C.c_void_p), packed.shape[1], pk2.ctypes.data_as(C.c_void_p),
                           l2.ctypes.data_as(C.c_void_p), f2.ctypes.data_as(C.c_void_p), th)
    print(f"rk_pack_reads_host, {th} threads, outputs touched: {n / (time.perf_counter() - t0) / 1e6:.1f} Mreads/s")

def rate(fn, reps=3):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return n * reps / (time.perf_counter() - t0) / 1e6

print(f"rk_place_batch_packed, pageable: {rate(lambda: pp.processQueriesPacked(packed, fixed_len=150, out=reuse)):.1f} Mreads/s")
print(f"rk_place_batch (ASCII), pageable: {rate(lambda: pp.processQueries(seq, off, out=reuse)):.1f} Mreads/s")

# page-locked caller buffers (rk_host_alloc): the DMA reads / writes them directly, no staging copies
lib.rk_host_alloc.restype = C.c_void_p
def pinned(shape, dtype):
    nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
    p = lib.rk_host_alloc(nbytes)
    return np.ctypeslib.as_array((C.c_uint8 * nbytes).from_address(p)).view(dtype).reshape(shape)
preuse = ra.Placements(pinned(n, np.uint8), pinned((n, K), np.uint16), pinned((n, K), np.float32), pinned((n, K), np.float64), pinned(n, np.uint32), {})
ppacked = pinned(packed.shape, np.uint32); ppacked[:] = packed
print(f"rk_place_batch_packed, page-locked in + out: {rate(lambda: pp.processQueriesPacked(ppacked, fixed_len=150, out=preuse)):.1f} Mreads/s")
print(f"rk_place_batch_packed, pageable in, page-locked out: {rate(lambda: pp.processQueriesPacked(packed, fixed_len=150, out=preuse)):.1f} Mreads/s")
for cr in (1 << 18, 1 << 17, 1 << 18, 1 << 19, 1 << 20, 1 << 18):
    os.environ["RK_CHUNK_READS"] = str(cr)
    print(f"chunk {cr}: packed pageable {rate(lambda: pp.processQueriesPacked(packed, fixed_len=150, out=reuse)):.1f}  page-locked {rate(lambda: pp.processQueriesPacked(ppacked, fixed_len=150, out=preuse)):.1f} Mreads/s")
