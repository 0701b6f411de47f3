"""Developer measurement: PCIe-inclusive rates of the host entry points on C2 (pageable vs page-locked buffers, characters vs packed
records) and of the host packer by thread count.  Run with RK_LIB=rappas_amd/librappas_place_dev.so to sweep the thread split
(RK_STAGE_THREADS / RK_DRAIN_THREADS) and the chunk size (RK_CHUNK_READS); RK_HOST_TIMING=1 prints where the host threads wait."""
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
print(f"host: {os.cpu_count()} cpus visible, {len(os.sched_getaffinity(0))} usable", flush=True)
packed, lens, flags = pp.pack_reads_host(seq, off)
lib = _lib.load()
P = lambda a: a.ctypes.data_as(C.c_void_p)
for th in (1, 2, 4, 8, 12, 16, 24, 32):
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        lib.rk_pack_reads_host(db.handle, n, P(seq), P(off), packed.shape[1], P(packed), P(lens), P(flags), th)
        best = min(best, time.perf_counter() - t0)
    print(f"rk_pack_reads_host, {th:2d} threads: {n / best / 1e6:7.1f} Mreads/s ({n * 150 / best / 1e9:5.1f} GB/s of characters)", flush=True)

def rate(fn, reps=4):
    fn(); fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return n * reps / (time.perf_counter() - t0) / 1e6

print(f"rk_place_batch (characters), pageable in + out: {rate(lambda: pp.processQueries(seq, off, out=reuse)):.1f} Mreads/s", flush=True)
print(f"rk_place_batch_packed, pageable in + out:        {rate(lambda: pp.processQueriesPacked(packed, fixed_len=150, out=reuse)):.1f} Mreads/s", flush=True)
preuse = ra.Placements(ra.host_alloc(n, np.uint8), ra.host_alloc((n, K), np.uint16), ra.host_alloc((n, K), np.float32), ra.host_alloc((n, K), np.float64), ra.host_alloc(n, np.uint32), {})
ppacked = ra.host_alloc(packed.shape, np.uint32); ppacked[:] = packed
pseq = ra.host_alloc(seq.shape, np.uint8); pseq[:] = seq
print(f"rk_place_batch_packed, page-locked in + out:     {rate(lambda: pp.processQueriesPacked(ppacked, fixed_len=150, out=preuse)):.1f} Mreads/s", flush=True)
print(f"rk_place_batch (characters), page-locked in + out (DMA of the characters, packed on the device): {rate(lambda: pp.processQueries(pseq, off, out=preuse)):.1f} Mreads/s", flush=True)
print(f"rk_place_batch (characters), pageable in, page-locked out: {rate(lambda: pp.processQueries(seq, off, out=preuse)):.1f} Mreads/s", flush=True)
if "dev" in os.path.basename(_lib.lib_path()):
    for st, dr in ((4, 4), (8, 4), (8, 8), (12, 4), (10, 6), (12, 8), (16, 8), (16, 16)):
        os.environ["RK_STAGE_THREADS"], os.environ["RK_DRAIN_THREADS"] = str(st), str(dr)
        print(f"stage {st:2d} / drain {dr:2d} threads: characters {rate(lambda: pp.processQueries(seq, off, out=reuse)):.1f}  packed {rate(lambda: pp.processQueriesPacked(packed, fixed_len=150, out=reuse)):.1f} Mreads/s", flush=True)
    del os.environ["RK_STAGE_THREADS"], os.environ["RK_DRAIN_THREADS"]
    for cr in (1 << 17, 1 << 18, 1 << 19, 1 << 20):
        os.environ["RK_CHUNK_READS"] = str(cr)
        print(f"chunk {cr}: characters {rate(lambda: pp.processQueries(seq, off, out=reuse)):.1f}  packed {rate(lambda: pp.processQueriesPacked(packed, fixed_len=150, out=reuse)):.1f} Mreads/s", flush=True)
