"""GPU (-m gpu): what the library must NOT do to the process that hosts it (a JVM, PyTorch).  Round 3's tile-order pre-pass raised the
release threshold of the device's DEFAULT memory pool and allocated from it on the caller's stream; since round 4 the scratch of a
launch belongs to the rk_db handle (rk_engine.hip: launch_scratch).  Also: the staging threads of the host path are kept on the CPUs
next to the GPU -- the CALLER's thread is never moved -- and d_flags_in may be the same buffer as the output flags."""
import ctypes as C
import os

import numpy as np
import pytest

import rappas_amd as ra
from rappas_amd import synth
from oracle import oracle as O
from tests.util import compare_with_oracle

pytestmark = pytest.mark.gpu

HIP_MEMPOOL_ATTR_RELEASE_THRESHOLD, HIP_MEMPOOL_ATTR_RESERVED_MEM_CURRENT, HIP_MEMPOOL_ATTR_USED_MEM_CURRENT = 4, 5, 7


def _default_pool_state():
    hip = C.CDLL("libamdhip64.so")
    pool = C.c_void_p()
    assert hip.hipDeviceGetDefaultMemPool(C.byref(pool), 0) == 0
    out = []
    for attr in (HIP_MEMPOOL_ATTR_RELEASE_THRESHOLD, HIP_MEMPOOL_ATTR_RESERVED_MEM_CURRENT, HIP_MEMPOOL_ATTR_USED_MEM_CURRENT):
        v = C.c_uint64(0)
        assert hip.hipMemPoolGetAttribute(pool, attr, C.byref(v)) == 0
        out.append(v.value)
    return out


@pytest.mark.parametrize("n_branches", [999, 7999, 40001])
def test_default_memory_pool_and_caller_thread_are_left_alone(n_branches):
    """batches large enough for the tile-order pre-pass (>= 32 768 reads) through the dense, the sorted-stream and the hash-table
    kernel, device and host entry points: the default pool's release threshold, its reserved and used bytes and the calling thread's
    CPU affinity are what they were; results equal the oracle's on a slice"""
    import torch
    before, aff = _default_pool_state(), os.sched_getaffinity(0)
    sdb = synth.make_db(4, 8, n_branches, 40000, 520000, seed=n_branches)
    seq, off = synth.make_reads(4, 70000, 150, seed=4, amb_rate=0.0005)
    db = ra.PhyloKmerDB.from_synth(sdb)
    try:
        pp = ra.PlacementProcess(db)
        got = pp.processQueries(seq, off)                      # host path: staging threads, page-locked staging buffers
        packed, lens, flags = pp.pack_reads_host(seq, off)
        pk = torch.from_numpy(packed.view(np.int32)).cuda()
        out = pp.place_packed(pk, fixed_len=150)               # device path on torch's stream
        torch.cuda.synchronize()
        assert _default_pool_state() == before
        assert os.sched_getaffinity(0) == aff
        odb = O.OracleDB.from_synth(sdb)
        sl = slice(60000, 62000)
        ref = odb.place(seq[int(off[sl.start]):int(off[sl.stop])], off[sl.start:sl.stop + 1] - off[sl.start])
        part = ra.Placements(got.n_rows[sl], got.branch[sl], got.score[sl], got.lwr[sl], got.flags[sl], {})
        compare_with_oracle(part, ref, odb, seq[int(off[sl.start]):], off[sl.start:sl.stop + 1] - off[sl.start])
        clean = (flags & ra._lib.RK_FLAG_AMBIGUOUS) == 0
        assert np.array_equal(out["branch"].cpu().numpy().view(np.uint16)[clean], got.branch[clean])
    finally:
        db.close()


@pytest.mark.parametrize("n_branches", [7999, 40001])
def test_input_flags_may_alias_the_output_flags(n_branches, monkeypatch, dev_lib):
    """rk_place_packed_device with d_flags_in == d_out->flags (one flag array, in place): reads flagged BAD_CHAR / AMBIGUOUS /
    TOO_LONG keep their flags when the first kernel hands their tile to place_packed16w_kernel -- the hand-over marks live in the
    launch's scratch, not in the flag word of the tile's first read (round 3) -- most tiles are handed over here (a small table)"""
    import torch
    monkeypatch.setenv("RK_HASH_KEY_SLACK", "1500")
    sdb = synth.make_db(4, 8, n_branches, 40000, 520000, seed=n_branches + 7)
    seq, off = synth.make_reads(4, 6000, 150, seed=9, amb_rate=0.002, bad_rate=0.02, var_len=60)
    db = ra.PhyloKmerDB.from_synth(sdb)
    try:
        pp = ra.PlacementProcess(db)
        packed, lens, flags = pp.pack_reads_host(seq, off)
        pk = torch.from_numpy(packed.view(np.int32)).cuda()
        ln = torch.from_numpy(lens.view(np.int32)).cuda()
        apart = pp.place_packed(pk, lens=ln, flags_in=torch.from_numpy(flags.view(np.int32)).cuda())
        fl = torch.from_numpy(flags.view(np.int32)).cuda()
        out = {k: v.clone() for k, v in apart.items()}
        out["flags"] = fl                                      # the caller's one flag array: input and output
        same = pp.place_packed(pk, lens=ln, flags_in=fl, out=out)
        torch.cuda.synchronize()
        for f in ("n_rows", "branch", "flags"):
            assert torch.equal(same[f], apart[f]), f
        assert torch.equal(same["score"].view(torch.int32), apart["score"].view(torch.int32))
        bad = (flags & (ra._lib.RK_FLAG_BAD_CHAR | ra._lib.RK_FLAG_AMBIGUOUS)) != 0
        assert bad.sum() > 50 and np.array_equal(same["flags"].cpu().numpy().view(np.uint32)[bad] & flags[bad], flags[bad])
    finally:
        db.close()
