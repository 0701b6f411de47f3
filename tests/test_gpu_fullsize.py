"""GPU (-m gpu): BASELINE config 2 at full size (10^7-entry DB, 10^7 x 150 bp reads) through size-independent
properties, plus an oracle check on a slice of the same batch."""
import numpy as np
import pytest

import rappas_amd as ra
from rappas_amd import synth
from oracle import oracle as O
from tests.util import compare_with_oracle

pytestmark = pytest.mark.gpu
N = 10_000_000
R = 150


@pytest.fixture(scope="module")
def full():
    import torch
    sdb = synth.make_config_db("C2")
    db = ra.PhyloKmerDB.from_synth(sdb)
    pp = ra.PlacementProcess(db)
    wpr = db.packed_words(R)
    gen = torch.Generator(device="cuda")
    gen.manual_seed(1)
    packed = torch.randint(-2**31, 2**31, (N, wpr), dtype=torch.int64, device="cuda", generator=gen).to(torch.int32)
    packed[:, wpr - 1] &= (1 << (R * 2 - 32 * (wpr - 1))) - 1
    out = pp.place_packed(packed, fixed_len=R)
    torch.cuda.synchronize()
    yield sdb, db, pp, packed, out
    db.close()


def test_fullsize_invariants(full):
    import torch
    sdb, db, pp, packed, out = full
    n_rows, score, lwr, flags, branch = out["n_rows"], out["score"], out["lwr"], out["flags"], out["branch"]
    placed = (flags & 1) != 0
    assert int(placed.sum()) > 0.999 * N                      # uniform reads vs a 75 %-coverage table: essentially all hit
    assert bool(((n_rows > 0) == placed).all())
    assert bool((flags & ~1 == 0).all())                      # no bad / short / ambiguous in this batch
    K = score.shape[1]
    idx = torch.arange(K, device="cuda")[None, :]
    on = idx < n_rows[:, None]
    # rows best -> worse, LWR in (0, 1], at most 1 in total, first row carries the largest weight
    s = torch.where(on, score, torch.full_like(score, float("-inf")))
    assert bool((s[:, :-1] >= s[:, 1:]).all())
    assert bool(((lwr > 0) == on).all()) and bool((lwr <= 1.0 + 1e-12).all())
    assert bool((lwr.sum(1) <= 1.0 + 1e-9).all())
    assert bool((lwr[:, :-1] >= lwr[:, 1:]).all())
    assert bool((lwr[placed, 0] >= 1.0 / 7 - 1e-12).all())
    # keep-factor: every emitted row is within 0.01 of the best one
    assert bool((torch.where(on, lwr, lwr[:, :1]) >= 0.01 * lwr[:, :1] * (1 - 1e-12)).all())
    # branch ids valid and distinct inside a read; scores are finite and below 0
    b = branch.to(torch.int32) & 0xFFFF
    assert bool((b[on] >= 1).all()) and bool((b[on] < sdb.n_branches).all())
    bs = torch.where(on, b, -(idx + 1).expand_as(b)).sort(1).values
    assert bool((bs[:, 1:] != bs[:, :-1]).all())
    assert bool(torch.isfinite(score[on]).all()) and bool((score[on] < 0).all())


def test_fullsize_deterministic_and_order_independent(full):
    """Idempotence + permutation: placing the same reads again, or in another order, gives the same records."""
    import torch
    sdb, db, pp, packed, out = full
    again = pp.place_packed(packed, fixed_len=R)
    torch.cuda.synchronize()
    for k in out:
        assert bool((again[k].view(torch.uint8) == out[k].view(torch.uint8)).all()), k
    perm = torch.randperm(N, device="cuda", generator=torch.Generator(device="cuda").manual_seed(5))
    shuf = pp.place_packed(packed[perm].contiguous(), fixed_len=R)
    torch.cuda.synchronize()
    for k in out:
        assert bool((shuf[k].view(torch.uint8) == out[k][perm].view(torch.uint8)).all()), k
    # other lane-group widths and the hashed table give bit-identical records (checksum of checksums)
    def checksum(o):
        return [int(o[k].view(torch.uint8).to(torch.int64).sum().item()) for k in sorted(o)] + \
               [int((o["score"].view(torch.int32).to(torch.int64) * torch.arange(1, 8, device="cuda")).sum().item())]
    want = checksum(out)
    for lanes in (32, 64):
        db.set_lanes_per_read(lanes)
        o = pp.place_packed(packed, fixed_len=R)
        torch.cuda.synchronize()
        assert checksum(o) == want, lanes
    db.set_lanes_per_read(0)
    hdb = ra.PhyloKmerDB.from_synth(sdb, table_mode=ra.RK_TABLE_HASH)
    o = ra.PlacementProcess(hdb).place_packed(packed, fixed_len=R)
    torch.cuda.synchronize()
    assert checksum(o) == want
    hdb.close()


def test_fullsize_slices_match_oracle(full):
    """Oracle parity on slices taken from the start, middle and end of the 10^7-read batch."""
    from bench import unpack_to_ascii
    sdb, db, pp, packed, out = full
    odb = O.OracleDB.from_synth(sdb)
    for a in (0, N // 2 + 12345, N - 1500):
        sl = slice(a, a + 1500)
        seq, off = unpack_to_ascii(4, packed[sl].cpu().numpy().view(np.uint32), R)
        got = ra.Placements(out["n_rows"][sl].cpu().numpy(), out["branch"][sl].cpu().numpy().view(np.uint16),
                            out["score"][sl].cpu().numpy(), out["lwr"][sl].cpu().numpy(),
                            out["flags"][sl].cpu().numpy().view(np.uint32), {})
        compare_with_oracle(got, odb.place(seq, off), odb, seq, off)
