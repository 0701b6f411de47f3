"""Seeded synthetic phylo-kmer DBs and reads (BASELINE.md section 4 / SURVEY.md section 8(d)).

Only shapes matter to the kernels: n_branches = 2*leaves-1 (branch id 0 = root never carries entries),
keys = random subset of the k-mer code space, row length ~ geometric with the requested mean, branch ids a
contiguous window of the pre-order id range, scores v = T*u with u~U(0,1) float32 so that T <= v <= 0
(what the reference's DB build can emit, src/core/algos/WordExplorer_v3.java:120-137).
"""
import math
from dataclasses import dataclass

import numpy as np

DNA_LETTERS = np.frombuffer(b"ATCG", dtype=np.uint8)                    # state order of DNAStatesShifted.java:33
AA_LETTERS = np.frombuffer(b"RHKDESTNQCGPAILMFWYV", dtype=np.uint8)     # AAStates.java:23-28

CONFIGS = {
    # name: (alphabet, k, leaves, n_keys, n_entries, read_len, n_reads)
    "C1": (4, 8, 50, 49_152, 500_000, 150, 1_000),
    "C2": (4, 10, 500, 786_432, 10_000_000, 150, 10_000_000),
    "C4": (20, 5, 200, 320_000, 3_200_000, 100, 1_000_000),
    # not BASELINE configs: C2's database and reads on larger reference trees (bench.py --config T4k / T8k / T20k / T64k; the
    # tree-size curve of scripts/tree_size_sweep.py as bench lines)
    "T4k": (4, 10, 2_000, 786_432, 10_000_000, 150, 4_000_000),
    "T8k": (4, 10, 4_000, 786_432, 10_000_000, 150, 4_000_000),
    "T20k": (4, 10, 10_000, 786_432, 10_000_000, 150, 2_000_000),
    "T40k": (4, 10, 20_000, 786_432, 10_000_000, 150, 2_000_000),  # (39 999 branches: between the two crossings of the short-row kernels)
    "T64k": (4, 10, 32_768, 786_432, 10_000_000, 150, 2_000_000),
    # a protein tree of 19 999 branches with C4-like rows (a quarter of the 5-mers present): scripts/hash_crossover.py aa as a bench line
    "P20k": (20, 5, 10_000, 786_432, 10_000_000, 100, 2_000_000),
    # not BASELINE configs either: long rows on trees between the dense kernels' and the workgroup-per-read kernel's regimes (a quarter
    # of the 9-mers present; rows of 400 entries on 9 001 branches, of 1 000 on 15 999: scripts/long_rows_big_tree.py as bench lines)
    "L9k": (4, 9, 4_501, 65_536, 26_214_400, 150, 300_000),
    "L16k": (4, 9, 8_000, 65_536, 65_536_000, 150, 300_000),
    "C5mini": (4, 12, 10_000, 1_048_576, 50_000_000, 250, 100_000),  # C5's tree/k at a DB size one test can hold
    # C5-shaped per-read work (19 999 branches, rows of ~2 600 entries, 250 bp => H ~ 4.7e5 entries/read) with a smaller
    # key space (k=8) so that the DB is ~1 GB instead of 200 GB: beyond the Infinity Cache, i.e. HBM-bound like C5
    "C5s": (4, 8, 10_000, 49_152, 127_795_200, 250, 200_000),
    # the same per-read work against a 12 GB database (k=10 at C2's key coverage, rows of ~2 600): 64-bit row offsets
    "C5m": (4, 10, 10_000, 786_432, 2_044_723_200, 250, 100_000),
}


def thresholds(omega, n_states, k):
    """float32 pair (PPStarThreshold, log10) as src/main_v2/Main_DBBUILD_3.java:165-166 computes it."""
    ratio = np.float32(omega) / np.float32(n_states)
    p = np.float32(math.pow(0.0 + float(ratio), k))
    t = np.float32(math.log10(float(p)))
    return p, t


@dataclass
class SynthDB:
    alphabet: int
    k: int
    n_branches: int
    thr: np.float32
    thr_log10: np.float32
    key_codes: np.ndarray    # u64 [n_keys]
    row_offsets: np.ndarray  # u64 [n_keys+1]
    branch_ids: np.ndarray   # u16 [n_entries]
    scores: np.ndarray       # f32 [n_entries]
    seed: int = 0

    @property
    def n_keys(self):
        return int(self.key_codes.shape[0])

    @property
    def n_entries(self):
        return int(self.branch_ids.shape[0])

    @property
    def bits(self):
        return 2 if self.alphabet == 4 else 5


def dense_to_code(alphabet, k, dense):
    """dense index in [0, sigma^k) -> key code (DNA: identical; AA: base-20 digits repacked 5 bits each)."""
    dense = np.asarray(dense, dtype=np.uint64)
    if alphabet == 4:
        return dense
    code = np.zeros_like(dense)
    rem = dense.copy()
    for i in range(k):
        code |= (rem % np.uint64(20)) << np.uint64(5 * i)
        rem //= np.uint64(20)
    return code


def make_db(alphabet, k, n_branches, n_keys, n_entries, seed=42, omega=1.5, sort_keys=False):
    rng = np.random.default_rng(seed)
    space = alphabet ** k
    n_keys = min(n_keys, space)
    dense = rng.choice(space, size=n_keys, replace=False) if n_keys < space else rng.permutation(space)
    if sort_keys:
        dense = np.sort(dense)
    key_codes = dense_to_code(alphabet, k, dense.astype(np.uint64))
    mean = max(1.0, n_entries / max(1, n_keys))
    lens = rng.geometric(1.0 / mean, size=n_keys).astype(np.int64)  # support {1,2,...}, mean `mean`
    np.minimum(lens, max(1, n_branches - 1), out=lens)
    row_offsets = np.zeros(n_keys + 1, dtype=np.uint64)
    np.cumsum(lens, out=row_offsets[1:])
    total = int(row_offsets[-1])
    # contiguous window [b0, b0+len) inside [1, n_branches)
    hi = np.maximum(1, n_branches - lens)  # b0 in [1, hi]
    b0 = 1 + (rng.random(n_keys) * hi).astype(np.int64)
    np.minimum(b0, hi, out=b0)
    if n_branches == 1:
        b0[:] = 0
    within = np.arange(total, dtype=np.int64) - np.repeat(row_offsets[:-1].astype(np.int64), lens)
    branch_ids = (np.repeat(b0, lens) + within).astype(np.uint16)
    thr, thr_log10 = thresholds(omega, alphabet, k)
    scores = (thr_log10 * rng.random(total, dtype=np.float32)).astype(np.float32)
    return SynthDB(alphabet, k, n_branches, thr, thr_log10, key_codes, row_offsets, branch_ids, scores, seed)


# ----------------------------------------------------------------------------------------------------------------------
# Counter-based generator: the host twin of rk_db_create_synth (rappas_amd/csrc/rk_synth_impl.h states the definition).
# The engine fills the HBM image from (seed, dense k-mer index, entry index) on the device; the functions below regenerate
# the same rows with numpy so that a checker can hand the rows a read touches to the oracle without the database ever
# existing on the host (C5: ~200 GB).
# ----------------------------------------------------------------------------------------------------------------------
_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _mix(x):
    x = np.asarray(x, dtype=np.uint64).copy()
    x ^= x >> np.uint64(30)
    x *= np.uint64(0xBF58476D1CE4E5B9)
    x ^= x >> np.uint64(27)
    x *= np.uint64(0x94D049BB133111EB)
    x ^= x >> np.uint64(31)
    return x


@dataclass
class SynthSpec:
    alphabet: int
    k: int
    n_branches: int
    thr: np.float32
    thr_log10: np.float32
    seed: int
    key_fraction: float
    mean_row_len: float

    @property
    def bits(self):
        return 2 if self.alphabet == 4 else 5

    @property
    def space(self):
        return self.alphabet ** self.k

    @property
    def max_len(self):
        return max(1, self.n_branches - 1)

    def surv(self):
        """surv[0] = 2^32, surv[j] = (surv[j-1] * q) >> 32 with q = floor((1 - 1/mean) * 2^32): P(len > j) in 32-bit fixed point."""
        if getattr(self, "_surv", None) is None:
            q = int((1.0 - 1.0 / float(self.mean_row_len)) * 4294967296.0)
            out = [1 << 32]
            while len(out) < self.max_len and out[-1] != 0:
                out.append((out[-1] * q) >> 32)
            self._surv = np.array(out, dtype=np.uint64)
        return self._surv

    def h0(self, dense):
        with np.errstate(over="ignore"):
            return _mix(np.uint64(self.seed & 0xFFFFFFFFFFFFFFFF) + (np.asarray(dense, dtype=np.uint64) + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15))

    def present_and_lens(self, dense):
        """(present bool[n], len int64[n]) for dense k-mer indices; len is meaningful where present."""
        h0 = self.h0(dense)
        thresh = np.uint64(int(float(self.key_fraction) * 4294967296.0))
        present = (h0 >> np.uint64(32)) < thresh
        r = _mix(h0 ^ np.uint64(0xA0761D6478BD642F)) >> np.uint64(32)
        sv = self.surv()
        # first j in [1, n) with surv[j] <= r  ==  1 + #{j in [1, n): surv[j] > r}; surv is non-increasing
        tail = sv[1:][::-1]  # ascending
        cnt = len(tail) - np.searchsorted(tail, r, side="right")
        lens = np.minimum(1 + cnt, self.max_len).astype(np.int64)
        return present, lens

    def rows(self, dense):
        """CSR rows (row_offsets u64[n+1], branch_ids u16, scores f32) of the given PRESENT dense indices, in the given order."""
        dense = np.asarray(dense, dtype=np.uint64)
        h0 = self.h0(dense)
        _, lens = self.present_and_lens(dense)
        n = len(dense)
        off = np.zeros(n + 1, dtype=np.uint64)
        np.cumsum(lens, out=off[1:])
        total = int(off[-1])
        if self.n_branches == 1:
            b0 = np.zeros(n, dtype=np.int64)
        else:
            hi = np.maximum(1, self.n_branches - lens).astype(np.uint64)
            b0 = 1 + (((_mix(h0 ^ np.uint64(0xE7037ED1A0B428DB)) >> np.uint64(32)) * hi) >> np.uint64(32)).astype(np.int64)
        branch = np.empty(total, dtype=np.uint16)
        scores = np.empty(total, dtype=np.float32)
        # in slices of ~3e7 entries: the temporaries are 8-byte arrays, the result 6 bytes per entry
        a = 0
        while a < n:
            b = int(np.searchsorted(off, off[a] + np.uint64(30_000_000), side="right"))
            b = min(n, max(a + 1, b - 1))
            lo, hi_e = int(off[a]), int(off[b])
            ln = lens[a:b]
            within = np.arange(hi_e - lo, dtype=np.int64) - np.repeat((off[a:b] - off[a]).astype(np.int64), ln)
            branch[lo:hi_e] = np.repeat(b0[a:b], ln) + within
            with np.errstate(over="ignore"):
                hs = _mix(np.repeat(h0[a:b], ln) + (within.astype(np.uint64) + np.uint64(1)) * np.uint64(0xD6E8FEB86659FD93))
            u = (hs >> np.uint64(40)).astype(np.float32) * np.float32(2.0 ** -24)
            scores[lo:hi_e] = np.float32(self.thr_log10) * u
            a = b
        return off, branch, scores

    def subset_db(self, dense):
        """SynthDB holding exactly the rows of the given dense indices that are present (for the oracle / rk_db_create)."""
        dense = np.unique(np.asarray(dense, dtype=np.uint64))
        present, _ = self.present_and_lens(dense)
        dense = dense[present]
        off, branch, scores = self.rows(dense)
        return SynthDB(self.alphabet, self.k, self.n_branches, self.thr, self.thr_log10,
                       dense_to_code(self.alphabet, self.k, dense), off, branch, scores, self.seed)

    def row_length_table(self):
        """int32[sigma^k]: row length per dense index, 0 where absent (for exact entry counts H)."""
        out = np.zeros(self.space, dtype=np.int32)
        step = 1 << 22
        for a in range(0, self.space, step):
            d = np.arange(a, min(self.space, a + step), dtype=np.uint64)
            p, ln = self.present_and_lens(d)
            out[a:a + len(d)] = np.where(p, ln, 0)
        return out


# name: (alphabet, k, leaves, key_fraction, mean_row_len, read_len, reads per GPU)
SPEC_CONFIGS = {
    # BASELINE C5: 10k-leaf DNA tree, k=12, ~3.3e10 entries (6 B each ~ 198 GB) -- 12 582 912 keys x rows of ~2 622
    "C5": (4, 12, 10_000, 0.75, 3.3e10 / 12_582_912, 250, 12_500_000),
    # the same generator at sizes tests can hold
    "C5tiny": (4, 8, 10_000, 0.75, 400.0, 250, 2_000),
}


def make_spec(name, seed=42, omega=1.5, mean_row_len=None, key_fraction=None):
    alphabet, k, leaves, kf, mean, _, _ = SPEC_CONFIGS[name]
    thr, thr_log10 = thresholds(omega, alphabet, k)
    return SynthSpec(alphabet, k, 2 * leaves - 1, thr, thr_log10, seed, key_fraction or kf, mean_row_len or mean)


def codes_of_reads(alphabet, k, seq, off):
    """dense k-mer indices of every window of every (unambiguous, equal-length or ragged) ASCII read -> u64 array (flat)."""
    letters = DNA_LETTERS if alphabet == 4 else AA_LETTERS
    table = np.full(256, 255, dtype=np.uint8)
    for i, ch in enumerate(letters):
        table[ch] = i
        table[ch + 32] = i
    out = []
    for r in range(len(off) - 1):
        st = table[seq[int(off[r]):int(off[r + 1])]].astype(np.uint64)
        if len(st) < k or (st == 255).any():
            continue
        Q = len(st) - k + 1
        idx = np.zeros(Q, dtype=np.uint64)
        for i in range(k):
            idx += st[i:i + Q] * np.uint64(alphabet ** i)
        out.append(idx)
    return np.concatenate(out) if out else np.zeros(0, dtype=np.uint64)


def make_motif_db(k, n_branches, genome_len=400, n_variants=3, mean_row=6, seed=5, omega=1.5, alphabet=4):
    """Database for k-mer spaces far too large to sample uniformly (DNA k >= 16, amino acids k >= 6): the keys are the k-mers
    of a random "genome" plus, for each, a few one- and two-symbol variants (so that the alternatives of ambiguity codes hit
    too).  Returns (SynthDB, genome str); reads for it are substrings of the genome (make_motif_reads)."""
    rng = np.random.default_rng(seed)
    g = rng.integers(0, alphabet, size=genome_len)
    bits = 2 if alphabet == 4 else 5
    codes = set()
    for j in range(genome_len - k + 1):
        w = g[j:j + k].copy()
        codes.add(int(sum(int(b) << (bits * i) for i, b in enumerate(w))))
        for _ in range(n_variants):
            v = w.copy()
            for p in rng.choice(k, size=int(rng.integers(1, 3)), replace=False):
                v[p] = rng.integers(0, alphabet)
            codes.add(int(sum(int(b) << (bits * i) for i, b in enumerate(v))))
    key_codes = np.array(sorted(codes), dtype=np.uint64)
    rng.shuffle(key_codes)
    n_keys = len(key_codes)
    lens = np.minimum(rng.geometric(1.0 / mean_row, size=n_keys), max(1, n_branches - 1)).astype(np.int64)
    off = np.zeros(n_keys + 1, dtype=np.uint64)
    np.cumsum(lens, out=off[1:])
    hi = np.maximum(1, n_branches - lens)
    b0 = np.minimum(1 + (rng.random(n_keys) * hi).astype(np.int64), hi)
    total = int(off[-1])
    within = np.arange(total, dtype=np.int64) - np.repeat(off[:-1].astype(np.int64), lens)
    branch = (np.repeat(b0, lens) + within).astype(np.uint16)
    thr, thr_log10 = thresholds(omega, alphabet, k)
    scores = (thr_log10 * rng.random(total, dtype=np.float32)).astype(np.float32)
    letters = (DNA_LETTERS if alphabet == 4 else AA_LETTERS).tobytes().decode()
    return SynthDB(alphabet, k, n_branches, thr, thr_log10, key_codes, off, branch, scores, seed), "".join(letters[b] for b in g)


def make_motif_reads(genome, n_reads, length, seed=1, amb_rate=0.0, var_len=0):
    """substrings of `genome`; with amb_rate, bases are replaced by an ambiguity code that CONTAINS the base (so the original
    k-mer is among the alternatives).  -> (seq uint8, off uint64)"""
    rng = np.random.default_rng(seed)
    if set(genome) <= set("ATCG"):
        containing = {"A": "RWMDHVN", "T": "YWKBDHN", "C": "YSMBHVN", "G": "RSKBDVN"}
    else:  # amino acids: X / * / - stand for all twenty, B = D|N, Z = E|Q, J = I|L
        containing = {c: "X*-x" for c in AA_LETTERS.tobytes().decode()}
        for amb, members in (("B", "DN"), ("Z", "EQ"), ("J", "IL")):
            for c in members:
                containing[c] += amb * 4
    reads = []
    for _ in range(n_reads):
        L = length - int(rng.integers(0, var_len + 1)) if var_len else length
        a = int(rng.integers(0, max(1, len(genome) - L + 1)))
        r = list(genome[a:a + L])
        if amb_rate > 0:
            for i in np.nonzero(rng.random(len(r)) < amb_rate)[0]:
                r[i] = containing[r[i]][int(rng.integers(0, len(containing[r[i]])))]
        reads.append("".join(r))
    off = np.zeros(n_reads + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(r) for r in reads])
    return np.frombuffer("".join(reads).encode(), dtype=np.uint8).copy(), off


def make_clade_db(k=10, n_branches=999, genome_len=700_000, mean_row=12.7, seed=5, omega=1.5, alphabet=4):
    """Database + genome for reads whose best branches are NEIGHBOURS (a clade: the usual shape of real placements).  The keys are
    the k-mers of a random genome; the rows of one 500-bp stretch cover the same few dozen branches, so the rows of a read cut from
    the genome pile up on one neighbourhood of the tree.  Every k-mer of such a read is present (hit rate 1 against the 0.75 of the
    uniform benchmark reads).  -> (SynthDB, genome states uint64[genome_len])"""
    rng = np.random.default_rng(seed)
    g = rng.integers(0, alphabet, size=genome_len).astype(np.uint64)
    codes = np.zeros(genome_len - k + 1, dtype=np.uint64)
    bits = 2 if alphabet == 4 else 5
    for i in range(k):
        codes += g[i:genome_len - k + 1 + i] << np.uint64(bits * i)
    key_codes, first = np.unique(codes, return_index=True)
    order = rng.permutation(len(key_codes))
    key_codes, pos = key_codes[order], first[order]
    nb = n_branches
    lens = np.minimum(rng.geometric(1.0 / mean_row, size=len(key_codes)), nb - 1).astype(np.int64)
    off = np.zeros(len(key_codes) + 1, dtype=np.uint64)
    np.cumsum(lens, out=off[1:])
    hi = np.maximum(1, nb - lens)
    block = pos // 500  # 500-bp stretches of the genome share a neighbourhood of the tree
    b0 = np.clip(1 + block * np.maximum(1, hi - 40) // (genome_len // 500 + 1) + rng.integers(-6, 7, size=len(pos)), 1, hi)
    total = int(off[-1])
    within = np.arange(total, dtype=np.int64) - np.repeat(off[:-1].astype(np.int64), lens)
    branch = (np.repeat(b0, lens) + within).astype(np.uint16)
    thr, thr_log10 = thresholds(omega, alphabet, k)
    scores = (thr_log10 * rng.random(total, dtype=np.float32)).astype(np.float32)
    return SynthDB(alphabet, k, nb, thr, thr_log10, key_codes, off, branch, scores, seed), g


def make_clade_reads(genome, n_reads, length=150, seed=6, alphabet=4):
    """substrings of the genome of make_clade_db -> (seq uint8 [n * length], off uint64 [n + 1])"""
    rng = np.random.default_rng(seed)
    starts = rng.integers(0, len(genome) - length, size=n_reads)
    idx = (starts[:, None] + np.arange(length)[None, :]).reshape(-1)
    seq = (DNA_LETTERS if alphabet == 4 else AA_LETTERS)[genome[idx].astype(np.int64)]
    return np.ascontiguousarray(seq), (np.arange(n_reads + 1, dtype=np.uint64) * np.uint64(length))


def make_config_db(name, seed=42, scale=1.0):
    alphabet, k, leaves, n_keys, n_entries, _, _ = CONFIGS[name]
    return make_db(alphabet, k, 2 * leaves - 1, int(n_keys * scale) if scale != 1.0 else n_keys,
                   int(n_entries * scale), seed=seed)


def make_reads(alphabet, n_reads, length, seed=1, amb_rate=0.0, bad_rate=0.0, var_len=0):
    """ASCII reads, uniform i.i.d. over the unambiguous alphabet (seed 1 like RandomSeqGenerator.java:20-21).

    amb_rate: per-symbol probability of an ambiguity character; bad_rate: per-read probability of one
    unsupported character; var_len: lengths uniform in [length-var_len, length].
    Returns (seq uint8 [total], off uint64 [n_reads+1]).
    """
    rng = np.random.default_rng(seed)
    letters = DNA_LETTERS if alphabet == 4 else AA_LETTERS
    if var_len:
        lens = rng.integers(max(0, length - var_len), length + 1, size=n_reads).astype(np.uint64)
    else:
        lens = np.full(n_reads, length, dtype=np.uint64)
    off = np.zeros(n_reads + 1, dtype=np.uint64)
    np.cumsum(lens, out=off[1:])
    total = int(off[-1])
    seq = letters[rng.integers(0, len(letters), size=total)]
    if amb_rate > 0:
        amb = np.frombuffer(b"NRYSWKMBDHVn-." if alphabet == 4 else b"XBZJ*-x", dtype=np.uint8)
        m = rng.random(total) < amb_rate
        seq = seq.copy()
        seq[m] = amb[rng.integers(0, len(amb), size=int(m.sum()))]
    if bad_rate > 0 and n_reads:
        seq = seq.copy()
        bad_reads = np.nonzero((rng.random(n_reads) < bad_rate) & (lens > 0))[0]
        pos = off[bad_reads] + (rng.random(len(bad_reads)) * lens[bad_reads]).astype(np.uint64)
        seq[pos.astype(np.int64)] = ord("@") if alphabet == 4 else ord("#")
    return np.ascontiguousarray(seq), off


def pack_reads_numpy(alphabet, seq, off, words_per_read=None):
    """Host-side reference packer for tests: symbol i at bits [i*B, (i+1)*B) of a little-endian bit string.
    Ambiguous / unsupported symbols pack as state 0 (the device packer does the same and flags the read)."""
    bits = 2 if alphabet == 4 else 5
    n = len(off) - 1
    lens = (off[1:] - off[:-1]).astype(np.int64)
    max_len = int(lens.max()) if n else 0
    wpr = words_per_read or max(1, (max_len * bits + 31) // 32)
    table = np.zeros(256, dtype=np.uint64)
    letters = DNA_LETTERS if alphabet == 4 else AA_LETTERS
    for i, ch in enumerate(letters):
        table[ch] = i
        table[ch + 32] = i
    if alphabet == 4:
        table[ord("U")] = table[ord("u")] = 1
    out = np.zeros((n, wpr), dtype=np.uint32)
    for r in range(n):
        s = table[seq[int(off[r]):int(off[r + 1])]]
        acc = 0
        for i, st in enumerate(s.tolist()):
            acc |= int(st) << (bits * i)
        for w in range(wpr):
            out[r, w] = (acc >> (32 * w)) & 0xFFFFFFFF
    return out, lens.astype(np.uint32)


def make_newick(n_nodes, seed=7):
    """Random tree with exactly `n_nodes` nodes for the host-side tools and their tests: binary and rooted when n_nodes is
    odd, an unrooted top level (three sons) when it is even.  Leaves are named t<id>, internal nodes n<id> (ids = order of
    appearance, as the reference's NewickReader numbers them), branch lengths are short decimals."""
    if n_nodes < 3:
        raise ValueError("need at least 3 nodes")
    rng = np.random.default_rng(seed)
    counter = [0]

    def split_odd(total, parts):
        # `parts` odd sizes summing to `total`
        left = total - parts
        out = [1] * parts
        for _ in range(left // 2):
            out[int(rng.integers(parts))] += 2
        return out

    def build(count, top):
        me = counter[0]
        counter[0] += 1
        bl = "" if top else ":%.4f" % (0.001 + float(rng.random()) * 0.5)
        if count == 1:
            return f"t{me}{bl}"
        sons = 3 if (count - 1) % 2 else 2
        sizes = split_odd(count - 1, sons)
        return "(" + ",".join(build(c, False) for c in sizes) + f")n{me}{bl}"

    import sys
    old = sys.getrecursionlimit()
    sys.setrecursionlimit(max(old, 4 * n_nodes + 100))
    try:
        return build(n_nodes, True) + ";"
    finally:
        sys.setrecursionlimit(old)


def make_pp_tables(alphabet, n_nodes, n_sites, seed=3, peaked=0.8, n_branches=None):
    """Synthetic ancestral-reconstruction output in the layout of the reference's PProbasSorted (core/PProbasSorted.java:19-25):
    states u8 / pp f32 [n_nodes, n_sites, alphabet], per site the states ranked by descending posterior and pp = log10 of it.
    `peaked` of the sites carry one dominant state (p in [0.9, 0.9995]), the others are diffuse, as posteriors of conserved /
    variable columns are.  Also returns node_branch u16[n_nodes]: the original branch a tested node's k-mers are filed under
    (two consecutive nodes share a branch, like the ghost nodes the reference hangs on every edge)."""
    rng = np.random.default_rng(seed)
    shape = (n_nodes, n_sites, alphabet)
    p = rng.random(shape) ** 3 + 1e-4
    dom = rng.random((n_nodes, n_sites)) < peaked
    top = 0.9 + 0.0995 * rng.random((n_nodes, n_sites))
    which = rng.integers(0, alphabet, (n_nodes, n_sites))
    onehot = np.zeros(shape, bool)
    np.put_along_axis(onehot, which[..., None], True, axis=2)
    rest = np.where(onehot, 0.0, p)
    rest = rest / rest.sum(axis=2, keepdims=True)
    peaked_p = np.where(onehot, top[..., None], rest * (1.0 - top[..., None]))
    diffuse_p = p / p.sum(axis=2, keepdims=True)
    prob = np.where(dom[..., None], peaked_p, diffuse_p).astype(np.float32)
    order = np.argsort(-prob, axis=2, kind="stable")
    states = order.astype(np.uint8)
    pp = np.log10(np.take_along_axis(prob, order, axis=2).astype(np.float64)).astype(np.float32)
    nb = n_branches if n_branches is not None else max(1, (n_nodes + 1) // 2)
    node_branch = (np.arange(n_nodes) // 2 % nb).astype(np.uint16)
    return states, pp, node_branch


def gap_intervals(rows):
    """CSR form (gap_off u32[L+1], gap_len i32[...]) of Alignment.getGapIntervals() (src/alignement/Alignment.java:232-258):
    for every alignment row, every maximal run of '-' contributes its length to the list of the site it starts at, each
    length once per site, in order of first appearance.  A run that reaches the end of the row is never closed by a residue and
    is not registered (updateGapIntervals only files a run when the next non-gap character is met)."""
    L = len(rows[0])
    lists = [[] for _ in range(L)]
    for r in rows:
        i = 0
        while i < L:
            if r[i] == "-":
                j = i
                while j < L and r[j] == "-":
                    j += 1
                if j < L and (j - i) not in lists[i]:
                    lists[i].append(j - i)
                i = j
            else:
                i += 1
    off = np.zeros(L + 1, np.uint32)
    off[1:] = np.cumsum([len(x) for x in lists])
    lens = np.array([v for x in lists for v in x], np.int32)
    return off, lens
