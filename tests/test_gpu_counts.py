"""rk_count_work_device (round 4): k-mers probed / k-mers with a row / row entries walked for a batch of packed reads -- the counts
the reference's loop would make (sk.getMerCount(), AmbigSequenceKnife.java:191; hash.getPairsOfTopPosition2(word) != null,
PlacementProcess.java:705-707; the pairs of :719-735) -- against a plain numpy count over the same reads, on every image layout."""
import numpy as np
import pytest

import rappas_amd as ra
from rappas_amd import synth

pytestmark = pytest.mark.gpu


def numpy_counts(sdb, alphabet, k, seq, off, flags):
    """reads flagged BAD_CHAR / TOO_LONG / AMBIGUOUS and reads shorter than k count nothing (the packed kernels skip them)"""
    keys = np.asarray(sdb.key_codes, dtype=np.uint64)
    bits = 2 if alphabet == 4 else 5
    dense = np.zeros(len(keys), dtype=np.uint64)
    for i in range(k):
        dense += ((keys >> np.uint64(bits * i)) & np.uint64((1 << bits) - 1)) * np.uint64(alphabet ** i)
    row_len = dict(zip(dense.tolist(), np.diff(np.asarray(sdb.row_offsets, dtype=np.int64)).tolist()))
    probed = hit = entries = 0
    skip = ra.RK_FLAG_BAD_CHAR | ra.RK_FLAG_TOO_LONG | ra.RK_FLAG_AMBIGUOUS
    for r in range(len(off) - 1):
        if flags[r] & skip or int(off[r + 1] - off[r]) < k:
            continue
        idx = synth.codes_of_reads(alphabet, k, seq[int(off[r]):int(off[r + 1])], np.array([0, int(off[r + 1] - off[r])], dtype=np.uint64))
        probed += len(idx)
        for c in idx.tolist():
            n = row_len.get(c)
            if n is not None:
                hit += 1
                entries += n
    return {"kmers_probed": probed, "kmers_hit": hit, "entries": entries}


SHAPES = {
    "dna_compact": lambda: (4, synth.make_db(4, 8, 301, 30000, 250000, seed=3)),
    "dna_long_rows_8_byte_table": lambda: (4, synth.make_db(4, 7, 9001, 3000, 3000 * 5000, seed=4)),   # rows beyond 4 080 entries
    "dna_windowed": lambda: (4, synth.make_db(4, 8, 7999, 30000, 400000, seed=5)),
    "dna_large_tree_image": lambda: (4, synth.make_db(4, 7, 19999, 4000, 4000 * 700, seed=6)),         # sorted 6-byte rows + index lines
    "dna_hashed_k16": lambda: (4, synth.make_motif_db(16, 499, genome_len=3000, seed=7)[0]),
    "aa_compact": lambda: (20, synth.make_db(20, 4, 399, 60000, 500000, seed=8)),
    "aa_hashed_k7": lambda: (20, synth.make_motif_db(7, 299, genome_len=2500, seed=9, alphabet=20)[0]),
}


@pytest.mark.parametrize("shape", sorted(SHAPES))
def test_work_counts_equal_a_numpy_count(shape):
    import torch
    alphabet, sdb = SHAPES[shape]()
    k = sdb.k
    if "hashed" in shape:  # reads cut from the genome the keys come from, so that k-mers hit
        genome = synth.make_motif_db(16 if alphabet == 4 else 7, 499 if alphabet == 4 else 299, genome_len=3000 if alphabet == 4 else 2500,
                                     seed=7 if alphabet == 4 else 9, alphabet=alphabet)[1]
        seq, off = synth.make_motif_reads(genome, 400, 90, seed=2, amb_rate=0.002, var_len=80)
    else:
        seq, off = synth.make_reads(alphabet, 1500, 110, seed=11, amb_rate=0.002, bad_rate=0.01, var_len=108)
    db = ra.PhyloKmerDB.from_synth(sdb)
    try:
        pp = ra.PlacementProcess(db)
        packed, lens, flags = pp.pack_reads_host(seq, off)
        want = numpy_counts(sdb, alphabet, k, seq, off, flags)
        dev = lambda x: torch.from_numpy(x.view(np.int32)).cuda()
        got = pp.count_work(dev(packed), lens=dev(lens), flags_in=dev(flags))
        assert got == want, (db.kernel_name(), got, want)
        assert want["kmers_hit"] > 0 and want["entries"] >= want["kmers_hit"]
        # reads of one length, no flag array
        n_fixed = 257
        seq2, off2 = synth.make_reads(alphabet, n_fixed, 100, seed=12) if "hashed" not in shape else (None, None)
        if seq2 is not None:
            p2, l2, f2 = pp.pack_reads_host(seq2, off2)
            assert pp.count_work(dev(p2), fixed_len=100) == numpy_counts(sdb, alphabet, k, seq2, off2, f2)
        assert pp.count_work(dev(packed)[:0], fixed_len=0) == {"kmers_probed": 0, "kmers_hit": 0, "entries": 0}
    finally:
        db.close()
