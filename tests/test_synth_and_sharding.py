"""CPU: synthetic generators, host packer, and the N>1 host logic (world_size-2 gloo)."""
import os
import socket

import numpy as np
import pytest

from rappas_amd import sharding, synth
from rappas_amd.placement import Placements
from oracle import oracle as O


def test_make_db_shape_and_determinism():
    a = synth.make_config_db("C1")
    b = synth.make_config_db("C1")
    assert a.n_keys == 49152 and abs(a.n_entries - 500_000) < 20_000 and a.n_branches == 99
    for f in ("key_codes", "row_offsets", "branch_ids", "scores"):
        assert (getattr(a, f) == getattr(b, f)).all()
    assert len(np.unique(a.key_codes)) == a.n_keys and int(a.key_codes.max()) < 4 ** 8
    assert a.branch_ids.min() >= 1 and a.branch_ids.max() < 99          # root (0) never carries entries
    assert (a.scores <= 0).all() and (a.scores >= a.thr_log10).all()    # T <= v <= 0
    lens = np.diff(a.row_offsets.astype(np.int64))
    assert lens.min() >= 1 and lens.max() <= 98
    r = 1234                                                            # rows are ascending contiguous windows
    w = a.branch_ids[int(a.row_offsets[r]):int(a.row_offsets[r + 1])]
    assert (np.diff(w.astype(int)) == 1).all()
    aa = synth.make_db(20, 3, 30, 500, 2000, seed=1)
    digits = [(aa.key_codes >> np.uint64(5 * i)) & np.uint64(31) for i in range(3)]
    assert all((d < 20).all() for d in digits) and (aa.key_codes >> np.uint64(15) == 0).all()


def test_reads_and_numpy_packer():
    seq, off = synth.make_reads(4, 50, 37, seed=1, var_len=20)
    assert set(np.unique(seq)) <= set(b"ACGT") and len(off) == 51
    packed, lens = synth.pack_reads_numpy(4, seq, off)
    for r in (0, 7, 49):
        s = bytes(seq[int(off[r]):int(off[r + 1])]).decode()
        states = [{"A": 0, "T": 1, "C": 2, "G": 3}[c] for c in s]
        assert lens[r] == len(s)
        for j in range(0, max(0, len(s) - 9)):
            bit = 2 * j
            v = (int(packed[r, bit >> 5]) | (int(packed[r, (bit >> 5) + 1]) << 32 if (bit >> 5) + 1 < packed.shape[1] else 0)) >> (bit & 31)
            assert v & ((1 << 20) - 1) == O.kmer_code(4, states[j:j + 10])
    seq, off = synth.make_reads(20, 10, 31, seed=2)
    packed, _ = synth.pack_reads_numpy(20, seq, off)
    s = bytes(seq[:31]).decode()
    states = ["RHKDESTNQCGPAILMFWYV".index(c) for c in s]
    acc = sum(int(w) << (32 * i) for i, w in enumerate(packed[0]))
    assert (acc >> (5 * 4)) & ((1 << 25) - 1) == O.kmer_code(20, states[4:9])


def test_shard_range_partition():
    for n in (0, 1, 7, 8, 1000, 10 ** 7 + 3):
        for w in (1, 2, 3, 8):
            spans = [sharding.shard_range(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # every rank owns a contiguous read shard and the same (replicated) DB; results come back in input order.
        sdb = synth.make_db(4, 6, 31, 1500, 9000, seed=5)
        seq, off = synth.make_reads(4, 101, 60, seed=9, var_len=30)
        lseq, loff = sharding.shard_reads(seq, off, world, rank)
        odb = O.OracleDB.from_synth(sdb)  # stands in for the per-GPU engine: this test covers the host logic only
        r = odb.place(lseq, loff)
        local = Placements(r["n_rows"], r["branch"], r["score"], r["lwr"], r["flags"] & ~np.uint32(O.RO_FLAG_TIE),
                           {"reads": int(len(loff) - 1)})
        allp = sharding.gather_placements(local, dst=0)
        t = sharding.max_over_ranks(1.0 + rank)
        if rank == 0:
            full = odb.place(seq, off)
            ok = ((allp.n_rows == full["n_rows"]).all() and (allp.branch == full["branch"]).all()
                  and (allp.score.view(np.uint32) == full["score"].view(np.uint32)).all()
                  and (allp.lwr == full["lwr"]).all() and allp.counters["reads"] == 101 and t == float(world))
            q.put(bool(ok))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_shard_and_concat():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True
