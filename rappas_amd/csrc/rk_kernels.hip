// rk_kernels.hip -- gfx950 (MI355X / CDNA4) kernels of the phylo-kmer placement hot path.
//
// What the reference does per read (src/core/algos/PlacementProcess.java:645-1075):
//   knife (AmbigSequenceKnife.java:98-272) -> compressMer (DNAStatesShifted.java:115-143) -> hash lookup
//   (CustomHash_v4_FastUtil81.java:146-153) -> S[x] accumulate (:719-735) -> top-K (:396-451) -> LWR (:974-1025).
//
// MI355X mapping (DESIGN.md section "Kernels"):
//   * a wavefront (64 lanes) is split into 64/G lane groups; each group owns one read at a time and a private
//     score vector S[n_branches] (f32 bit patterns) + a hit list of row descriptors, both in LDS;
//   * probe phase: lane <-> k-mer position; the k-mer code is funnel-shifted out of the packed read in
//     registers; one 8-byte descriptor gather per k-mer (direct table) or a linear probe over 16-byte slots
//     (hash table); hits are compacted into the LDS list with a ballot + popcount;
//   * accumulate phase: rows are applied strictly in k-mer order (bit parity with the reference's sequential
//     float32 adds); inside a row (branches unique) the G lanes of the group each own one entry and do a plain
//     LDS read-modify-write -- no atomics, LDS operations of a wave execute in program order;  row gathers are
//     software-pipelined through a register ring of depth U so U row chunks are in flight per lane;
//   * select phase: per-lane sorted insertion of packed 64-bit keys (ordered score bits << 32 | ~branch),
//     K rounds of group arg-max, LWR in fp64 on the first K lanes; the scan also resets S for the next read.
// No MFMA: this is a gather/accumulate bounded by the memory system, not a contraction.
#include "rk_device.h"

#include <type_traits>
#ifndef RK_ROW_NT
#define RK_ROW_NT 0
#endif
#ifndef RK_ROW_AUX
#define RK_ROW_AUX 0  // cache-policy bits of the row-unit loads (developer A/B: 1 = sc0, 2 = nt, 16 = sc1)
#endif
#ifndef RK_HEADS_MAX_K
#define RK_HEADS_MAX_K 16  // the fast select (stream heads) up to this keep_at_most (C2: K = 9 ... 16 run at 343 ... 274 Mreads/s with it, 280 ... 117 through the exact two-pass scan)
#endif
#ifndef RK_ABLATE
#define RK_ABLATE 0  // timing-only dev builds: 1 = no accumulate, 2 = no select, 4 = no LWR, 8 = no LDS update, 16 = one cached row line, 32 = one unit per row, 128 / 256 = ambiguity kernel without amb_position / without its row accumulate, 512 / 1024 / 2048 = windowed kernel without accumulate / with the exact select only in the last window / without compaction and accumulate (outputs are then wrong), 4096 = windowed kernel without skipping untouched windows (outputs stay right), 8192 = no result stores
#endif
#include "../../include/rappas_place.h"

namespace rk {

#ifdef RK_STAMPS
// Diagnostic build only (scripts/stamps.py): per-wave cycle sums of the kernel phases. Never compiled into the product.
__device__ unsigned long long rk_stamp_buf[2 * 4096 * 16];  // second half: place_packed16w_kernel launched behind place_packed16s_kernel
__device__ __forceinline__ unsigned long long rk_now() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#define RK_STAMP(slot) do { unsigned long long n_ = rk_now(); st_[slot] += n_ - t_; t_ = n_; } while (0)
#define RK_STAMP_PARAMS , unsigned long long *st_, unsigned long long &t_
#define RK_STAMP_ARGS , st_, t_
#else
#define RK_STAMP(slot) do {} while (0)
#define RK_STAMP_PARAMS
#define RK_STAMP_ARGS
#endif

// ------------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ u32 ord_f32(float f) {  // monotone map float -> u32 (Float.compare order)
    u32 u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float unord_f32(u32 o) {
    u32 u = (o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o;
    return __uint_as_float(u);
}
__device__ __forceinline__ u64 make_key(u32 raw_score_bits, u32 branch) {
    return ((u64)ord_f32(__uint_as_float(raw_score_bits)) << 32) | (u64)(0xFFFFu - branch);
}
__device__ __forceinline__ u64 shfl64(u64 v, int src, int width) {
    u32 lo = (u32)v, hi = (u32)(v >> 32);
    lo = __shfl(lo, src, width);
    hi = __shfl(hi, src, width);
    return ((u64)hi << 32) | lo;
}
__device__ __forceinline__ u64 shfl_xor64(u64 v, int m, int width) {
    u32 lo = (u32)v, hi = (u32)(v >> 32);
    lo = __shfl_xor(lo, m, width);
    hi = __shfl_xor(hi, m, width);
    return ((u64)hi << 32) | lo;
}
__device__ __forceinline__ double shfl_f64(double v, int src, int width) {
    return __longlong_as_double((long long)shfl64((u64)__double_as_longlong(v), src, width));
}
// the read a tile's slot stands for: PlaceArgs::perm when the batch's reads were grouped by their place in the tree (retile_* below)
__device__ __forceinline__ bool tile_order_given(const PlaceArgs &a) { return a.perm != nullptr && *a.keep_order == 0u; }  // (once per wave)
// the verdicts of the pre-pass on the batch (retile_decide_kernel: keep_order[0] = the batch keeps its order, keep_order[3] = its sampled
// k-mers have a row no more often than a random read's) against what this launch was made for (PlaceArgs::only_if)
__device__ __forceinline__ bool batch_is_mine(const PlaceArgs &a) {
    if (!a.only_if || !a.perm) return true;
    const u32 cls = a.keep_order[0] == 0u ? 2u : (a.keep_order[3] != 0u ? 1u : 0u);
    return ((a.only_if >> cls) & 1u) != 0u;
}
__device__ __forceinline__ u64 tile_read(const PlaceArgs &a, u64 slot, bool given) { return given ? (u64)a.perm[slot] : slot; }
// LDS data exchanged between lanes of ONE wave: DS operations of a wave execute in order, so only the
// compiler has to be stopped from reordering / caching.
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// ---- wave ballots cut into lane groups, and counts below a lane, WITHOUT 64-bit shifts by a variable count: on gfx950 such a shift
//      reads its count from v0 when hipcc happens to put it into the kernel's last VGPR; the build refuses that code
//      (tools/check_isa.py), these forms do not produce it in the first place ----
// the G bits of a wave ballot that belong to lane group gi (G consecutive lanes)
template <int G>
__device__ __forceinline__ u64 group_bits(u64 bal, u32 gi) {
    if (G == 64) return bal;
    u32 lo = (u32)bal, hi = (u32)(bal >> 32);
    asm("" : "+v"(lo), "+v"(hi));  // (two separate words from here on: left to itself the optimiser puts the choice of a half back together as a 64-bit shift by gi * G)
    if (G == 32) return gi ? hi : lo;
    constexpr u32 PER = 32 / (G < 32 ? G : 32);  // groups per 32-bit half
    const u32 half = gi >= PER ? hi : lo;
    return (half >> ((gi % PER) * (u32)G)) & ((1u << (G < 32 ? G : 31)) - 1u);
}
// popcount(m & ((1 << li) - 1)) for a group's bits m (li = lane in group; G == 64: li is the lane itself)
template <int G>
__device__ __forceinline__ int count_below(u64 m, u32 li) {
    if (G <= 32) return __builtin_popcount((u32)m & ((1u << li) - 1u));
    return (int)__builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0u));
}
// the n lowest bits of a 64-bit word set, n in 0 .. 64 (a mask of windows / of lanes below a position), from two 32-bit words
__device__ __forceinline__ u64 bits_below(u32 n) {
    u32 lo = n >= 32u ? 0xFFFFFFFFu : ((1u << (n & 31u)) - 1u);
    u32 hi = n <= 32u ? 0u : (n >= 64u ? 0xFFFFFFFFu : ((1u << ((n - 32u) & 31u)) - 1u));
    asm("" : "+v"(lo), "+v"(hi));  // (kept apart: see group_bits)
    return ((u64)hi << 32) | lo;
}
// a small value (a state: < 32) moved to bit position sh < 64 of a 64-bit code
__device__ __forceinline__ u64 place_bits(u32 v, u32 sh) {
    // bits of v that stay in the low word / spill into the high word, by 32-bit shifts (v < 2^5: at most 4 bits spill)
    u32 lo = sh < 32u ? v << (sh & 31u) : 0u;
    u32 hi = sh >= 32u ? v << ((sh - 32u) & 31u) : ((sh & 31u) ? v >> ((32u - sh) & 31u) : 0u);
    asm("" : "+v"(lo), "+v"(hi));  // (kept apart: see group_bits)
    return ((u64)hi << 32) | lo;
}
// bit `lane` of a wave mask
__device__ __forceinline__ bool lane_bit(u64 m, u32 lane) {
    u32 lo = (u32)m, hi = (u32)(m >> 32);
    asm("" : "+v"(lo), "+v"(hi));
    return (((lane & 32u) ? hi : lo) >> (lane & 31u)) & 1u;
}
// bits [sh, sh + 64) of the 96-bit little-endian string w2:w1:w0 (sh < 32): two 32-bit funnel shifts (v_alignbit_b32)
__device__ __forceinline__ u64 funnel96(u32 w0, u32 w1, u32 w2, u32 sh) {
    return ((u64)__builtin_amdgcn_alignbit(w2, w1, sh) << 32) | __builtin_amdgcn_alignbit(w1, w0, sh);
}

// entry e of the row a descriptor points at, for either row layout (8-byte {branch, score} pairs, or -- large-tree
// images -- a u16 branch array followed by an f32 score array)
__device__ __forceinline__ void load_entry(const DbView &db, u64 desc, u32 e, u32 &br, float &sc) {
    const unsigned char *base = db.rows + ((desc >> DESC_LEN_BITS) << 3);
    if (db.soa) {
        const u32 len = (u32)desc & DESC_LEN_MASK;
        br = ((const unsigned short *)base)[e];
        sc = ((const float *)(base + 2 * (size_t)len))[e];
    } else {
        const Entry en = ((const Entry *)base)[e];
        br = en.branch ? (en.branch >> 2) - 1u : 0xFFFFu;  // slot offset -> branch id; all-zero padding -> "skip"
        sc = en.score;
    }
}

// k-mer code at symbol position j of a packed record (symbol i at bits [i*BITS, (i+1)*BITS) of the
// little-endian bit string).  DNA: this integer IS the reference key (compressMer bytes little-endian,
// DNAStatesShifted.java:115-143).  AA: sum state_i << 5i (compressMer = identity, AAStates.java:195-197).
template <int BITS>
__device__ __forceinline__ u64 extract_code(const u32 *rec, u32 words, u32 j, u32 k) {
    // all loads unconditional (indices clamped, results masked) so that s_waitcnt counting stays exact
    const u32 bit = j * BITS;
    const u32 wi = bit >> 5, sh = bit & 31;
    const u32 last = words - 1;
    const u32 w0 = rec[wi];
    const u32 i1 = wi + 1 < words ? wi + 1 : last;
    u32 w1 = rec[i1];
    w1 = wi + 1 < words ? w1 : 0u;
    const u32 nbits = k * BITS;
    u32 w2 = 0u;
    if (BITS == 5 || nbits + 31 > 64) {  // AA with k >= 7 and DNA with k >= 17 can need a third word (uniform test)
        const u32 i2 = wi + 2 < words ? wi + 2 : last;
        w2 = rec[i2];
        w2 = (wi + 2 < words && nbits + sh > 64) ? w2 : 0u;
    }
    const u64 code = funnel96(w0, w1, w2, sh);  // (32-bit funnel shifts: no 64-bit shift by a per-lane count)
    return nbits >= 64 ? code : (code & ((1ull << nbits) - 1));
}

// dense index of a code in the direct table: DNA -> the code itself; AA -> base-20 number of its 5-bit digits
template <int BITS>
__device__ __forceinline__ u64 dense_index(u64 code, u32 k) {
    if (BITS == 2) return code;
    if (k <= 7) {
        // direct tables exist for 20^k <= 2^31, i.e. k <= 7: digits 0..5 sit in the low word, digit 6 straddles it, digits at and
        // above k are zero -- Horner from the top in 32-bit arithmetic, two instructions a digit (the general loop below is 64-bit
        // shifts and multiplies with a run-time trip count: ~10 x the work, paid twice per probe)
        const u32 lo = (u32)code;
        u32 idx = (u32)(code >> 30) & 31u;
        idx = idx * 20u + ((lo >> 25) & 31u);
        idx = idx * 20u + ((lo >> 20) & 31u);
        idx = idx * 20u + ((lo >> 15) & 31u);
        idx = idx * 20u + ((lo >> 10) & 31u);
        idx = idx * 20u + ((lo >> 5) & 31u);
        idx = idx * 20u + (lo & 31u);
        return idx;
    }
    u64 idx = 0, pw = 1;
    for (u32 i = 0; i < k; i++) {
        idx += ((code >> (5 * i)) & 31) * pw;
        pw *= 20;
    }
    return idx;
}

// Table lookups are split into a FETCH (one unconditional gather per k-mer, so that a whole batch of them can be
// in flight) and a branch-free DECODE.  (Written as one function, hipcc sank part of the 16-byte load under a
// data-dependent branch and waited vmcnt(0) after every probe: the batch ran one gather at a time.)
struct RawSlot {
    uint4 v;
};

template <int BITS, int TM>
__device__ __forceinline__ RawSlot lookup_fetch(const DbView &db, u64 code) {
    RawSlot r;
    if (TM == TM_COMPACT) {
        const u32 idx = (u32)dense_index<BITS>(code, db.k);  // (direct tables: sigma^k <= 2^31 -- 32-bit division by a constant, not 64-bit)
        const u32 blk = idx / COMPACT_KMERS;
        r.v = db.compact[db.compact_nib ? blk >> 1 : blk];
    } else {  // TM_DIRECT8
        const uint2 d = *(const uint2 *)(db.direct + (u32)dense_index<BITS>(code, db.k));
        r.v = make_uint4(d.x, d.y, 0u, 0u);
    }
    return r;
}

template <int BITS, int TM>
__device__ __forceinline__ u64 lookup_decode(const DbView &db, const RawSlot &r, u64 code) {
    if (TM == TM_COMPACT) {
        // 16-byte block of COMPACT_KMERS (12) consecutive k-mers: {u32 first 128-byte unit of the block's rows,
        // 12 x u8 units per row}; the row offset is the block base plus a byte prefix sum (v_sad_u8 accumulates)
        const u32 idx = (u32)dense_index<BITS>(code, db.k);
        const u32 i = idx % COMPACT_KMERS;
        const uint4 n = r.v;
        if (db.compact_nib) {
            // the half-size form (no row of the database exceeds 15 units): {u32 first unit, 24 x u4 units per row}; k-mer i of the
            // odd 12-block sits in the block's second half.  Nibble sums: even and odd nibbles through v_sad_u8 each.
            const u32 j = i + ((idx / COMPACT_KMERS) & 1u) * COMPACT_KMERS;
            auto nsum = [](u32 w, u32 acc) {
                return __builtin_amdgcn_sad_u8(w & 0x0F0F0F0Fu, 0u, __builtin_amdgcn_sad_u8((w >> 4) & 0x0F0F0F0Fu, 0u, acc));
            };
            const u32 c0 = nsum(n.y, 0u);
            const u32 c1 = nsum(n.z, c0);
            const u32 word = j >> 3, sh = (j & 7u) * 4u;
            const u32 wsel = word == 0 ? n.y : (word == 1 ? n.z : n.w);
            const u32 csel = word == 0 ? 0u : (word == 1 ? c0 : c1);
            const u32 prefix = nsum(wsel & ((1u << sh) - 1u), csel);
            const u32 mine = (wsel >> sh) & 15u;
            const u64 unit = (u64)n.x + prefix;
            const u64 d = ((unit * ROW_UNIT) << DESC_LEN_BITS) | (u64)(mine * ROW_UNIT);
            return d & (0ull - (u64)(mine != 0));
        }
        const u32 c0 = __builtin_amdgcn_sad_u8(n.y, 0u, 0u);
        const u32 c1 = __builtin_amdgcn_sad_u8(n.z, 0u, c0);
        const u32 word = i >> 2, sh = (i & 3u) * 8u;
        const u32 wsel = word == 0 ? n.y : (word == 1 ? n.z : n.w);
        const u32 csel = word == 0 ? 0u : (word == 1 ? c0 : c1);
        const u32 prefix = __builtin_amdgcn_sad_u8(wsel & ((1u << sh) - 1u), 0u, csel);
        const u32 mine = (wsel >> sh) & 0xFFu;
        const u64 unit = (u64)n.x + prefix;
        const u64 d = ((unit * ROW_UNIT) << DESC_LEN_BITS) | (u64)(mine * ROW_UNIT);
        return d & (0ull - (u64)(mine != 0));
    } else {
        return ((u64)r.v.y << 32) | r.v.x;
    }
}

template <int BITS, int TM>
__device__ __forceinline__ u64 lookup_desc(const DbView &db, u64 code) {
    if (TM != TM_HASH) {
        const RawSlot r = lookup_fetch<BITS, TM>(db, code);
        return lookup_decode<BITS, TM>(db, r, code);
    } else {
        u64 h = mix64(code) & db.hash_mask;
        const u64 want = code + 1;
        while (true) {
            uint4 s = db.slots[h];
            u64 key = ((u64)s.y << 32) | s.x;
            if (key == want) return ((u64)s.w << 32) | s.z;
            if (key == 0) return 0;
            h = (h + 1) & db.hash_mask;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// accumulate: row cursor + register ring
// ------------------------------------------------------------------------------------------------
// Offsets into the row blob are 32-bit when the blob is < 4 GiB (WIDE=false: the loads use the
// scalar-base + 32-bit-vgpr-offset form, no 64-bit VALU address arithmetic), 64-bit otherwise.
template <bool WIDE> struct OffsetT { typedef u32 type; };
template <> struct OffsetT<true> { typedef u64 type; };

template <bool WIDE>
struct Cursor {
    typedef typename OffsetT<WIDE>::type off_t;
    int h;      // index of the current row in the hit list
    int rem;    // entries of the current row not yet issued (<= 0: fetch the next row)
    off_t cur;  // byte offset of the next chunk of the current row (SOA: of its branch ids)
    off_t scur; // SOA only: byte offset of the next chunk's scores
    u64 dn;     // list[min(h + 1, cnt)], fetched one step ahead so its LDS latency is off the critical path
};

// Branch-free on purpose: every lane always issues its row load (lanes without an entry read the reserved entry
// at blob offset 0, which updates the scratch slot), and nothing selects on a loaded value here, so the
// compiler counts outstanding loads exactly (s_waitcnt vmcnt(N)) and the register ring keeps U chunks in flight
// per lane.  list[cnt] must be a zero descriptor (sentinel).  SOA: rows are u16 branch[len] | f32 score[len].
template <int G, bool WIDE, bool SOA>
__device__ __forceinline__ void cursor_issue(Cursor<WIDE> &c, const u64 *list, int cnt, u32 li, int lii,
                                             const unsigned char *rows, u32 &br, float &sc, u32 s_lo = 0, u32 s_win = 0xFFFFu) {
    typedef typename OffsetT<WIDE>::type off_t;
    const bool need = c.rem <= 0;
    int hn = c.h + (need ? 1 : 0);
    hn = hn < cnt ? hn : cnt;
    const u64 d = c.dn;
    const int hp = hn + 1 < cnt ? hn + 1 : cnt;
    c.dn = list[hp];
    const u32 len = (u32)d & DESC_LEN_MASK;
    const off_t off = (off_t)(d >> DESC_LEN_BITS) << 3;
    c.h = hn;
    c.rem = need ? (int)len : c.rem;
    c.cur = need ? off : c.cur;
    if (SOA) c.scur = need ? (off_t)(off + 2 * (off_t)len) : c.scur;
    const bool ok = lii < c.rem;
    if (SOA) {
        const off_t bo = ok ? (off_t)(c.cur + 2 * li) : (off_t)0;
        const off_t so = ok ? (off_t)(c.scur + 4 * li) : (off_t)0;
        br = *(const unsigned short *)(rows + bo);  // raw id, turned into a slot offset at apply time (accumulate_list)
        sc = *(const float *)(rows + so);
        c.cur += 2 * G;
        c.scur += 4 * G;
    } else {
        off_t eo = ok ? (off_t)(c.cur + 8 * li) : (off_t)0;
        if (RK_ABLATE & 16) eo = 0;  // timing-only: every row load reads the reserved "skip" entry
        const uint2 e = *(const uint2 *)(rows + eo);
        br = e.x;  // (raw: the windowed kernel rebases it at apply time, see accumulate_list)
        sc = __uint_as_float(e.y);
        c.cur += 8 * G;
    }
    c.rem -= G;
}

// S[x] update of PlacementProcess.java:726-733: first touch seeds fl(Q*T), then S = fl(S + fl(v - T)).
// Branch-free: lanes without an entry (br == 0xFFFF) update the group's scratch word S[nb] instead, so the
// compiler can overlap the LDS read latency with the address arithmetic of the next ring slot.
__device__ __forceinline__ void apply_entry(u32 *S, u32 nb, u32 br, float sc, float QT, float T, u32 br_base = 0) {
    if (RK_ABLATE & 8) {  // timing-only: keep the loads alive, skip the LDS update
        asm volatile("" ::"v"(br), "v"(sc));
        return;
    }
    // (S holds the branches [br_base, br_base + nb) of the pass; word nb is the scratch slot)
    const u32 idx = (br != 0xFFFFu) ? br - br_base : nb;
    const u32 old = S[idx];
    const float base = (old == S_UNTOUCHED) ? QT : __uint_as_float(old);
    const float d = sc - T;
    const float nw = base + d;
    S[idx] = __float_as_uint(nw);
}
// Same update for the slot layout of the packed / ASCII kernels: word 0 of S is the scratch slot, branch x lives in
// word x + 1, and `sb` is the word's byte offset as the row entries carry it (0 for padding).
// MONO (every score of the database is >= the threshold, so every increment is >= 0 and a touched word never drops
// below Q*T): the first-touch test is one v_max with the -inf marker.
template <bool MONO = false>
__device__ __forceinline__ void apply_slot(u32 *S, u32 sb, float sc, float QT, float T) {
    if (RK_ABLATE & 8) {
        asm volatile("" ::"v"(sb), "v"(sc));
        return;
    }
    u32 *p = (u32 *)((unsigned char *)S + sb);
    const u32 old = *p;
    float base;
    if (MONO) asm("v_max_f32 %0, %1, %2" : "=v"(base) : "v"(old), "v"(QT));  // plain max: no NaNs here, no canonicalize
    else base = (old == S_UNTOUCHED) ? QT : __uint_as_float(old);
    const float d = sc - T;
    const float nw = base + d;
    *p = __float_as_uint(nw);
}
template <int G, int U, bool WIDE, bool SOA = false>
__device__ __forceinline__ void accumulate_list(u32 *S, u32 nb, const u64 *list, int cnt, u32 li,
                                                const unsigned char *rows, float QT, float T, u32 s_lo = 0, u32 s_win = 0xFFFFu,
                                                u32 wlo4p4 = 4u, u32 w4 = 0xFFFFFFFFu) {
    Cursor<WIDE> c;
    c.h = -1; c.rem = 0; c.cur = 0; c.scur = 0;
    c.dn = list[0];
    u32 br[U];
    float sc[U];
#pragma unroll
    for (int u = 0; u < U; u++) cursor_issue<G, WIDE, SOA>(c, list, cnt, li, (int)li, rows, br[u], sc[u], s_lo, s_win);
    // what a loaded entry's first word means is settled only when its step comes (anything computed from it at load time would
    // wait for the load and serialise the ring):
    //   large-tree images: a raw branch id -> slot offset inside the window [s_lo, s_lo + s_win) of the tree S holds (the whole
    //     tree unless it exceeds the LDS); 0xFFFF = padding / reserved line 0 and ids outside the window go to the scratch word;
    //   slot-offset images: already a slot offset; the windowed kernel rebases it into its window.
    const bool win = !SOA && w4 != 0xFFFFFFFFu;
    auto slot_of = [&](u32 b) {
        if (SOA) {
            const u32 xw = b - s_lo;
            return (xw < s_win) ? (xw + 1u) * 4u : 0u;
        }
        const u32 t = b - wlo4p4;
        return win ? ((t < w4) ? t + 4u : 0u) : b;
    };
    while (true) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            apply_slot(S, slot_of(br[u]), sc[u], QT, T);
            cursor_issue<G, WIDE, SOA>(c, list, cnt, li, (int)li, rows, br[u], sc[u], s_lo, s_win);
        }
        if (!__any(c.h < cnt)) break;
    }
#pragma unroll
    for (int u = 0; u < U; u++) apply_slot(S, slot_of(br[u]), sc[u], QT, T);
}

// ------------------------------------------------------------------------------------------------
// accumulate by chunks, general form (any group width, any blob size; G == 16 with 32-bit offsets takes
// accumulate_units below instead): the hit list holds one item per CHUNK (<= G entries of one row, whole 64-byte
// lines), written in k-mer order by the probe phase.  A step is then: decode one item, one entry load per lane, one
// LDS read-modify-write -- no row cursor.  Items are 32-bit when the row blob is < 4 GiB:
//   (64-byte line index of the chunk) << 4 | (lines in the chunk),   else the 64-bit descriptor format.
// Items are read one ring-iteration ahead so their LDS latency is hidden; the list region has 3U slots of slack.
// ------------------------------------------------------------------------------------------------
template <bool WIDE> struct ItemT { typedef u32 type; };
template <> struct ItemT<true> { typedef u64 type; };

template <int G, bool WIDE>
__device__ __forceinline__ typename ItemT<WIDE>::type make_item(u64 off8, u32 n) {
    if (WIDE) return (typename ItemT<WIDE>::type)((off8 << DESC_LEN_BITS) | n);
    return (typename ItemT<WIDE>::type)((u32)(off8 >> 3) << 4 | (n >> 3));
}

template <int G, bool WIDE>
__device__ __forceinline__ void chunk_issue(typename ItemT<WIDE>::type it, bool in, u32 li, u32 li8,
                                            const unsigned char *rows, u32 &br, float &sc) {
    typedef typename OffsetT<WIDE>::type off_t;
    bool ok;
    off_t eo;
    if (WIDE) {
        const u32 n = (u32)it & DESC_LEN_MASK;
        ok = in && li < n;
        eo = (off_t)(((u64)it >> DESC_LEN_BITS) << 3) + li8;
    } else {
        ok = in && (li >> 3) < ((u32)it & 15u);
        eo = (off_t)((((u32)it & ~15u) << 2) + li8);
    }
    eo = ok ? eo : (off_t)0;
    if (RK_ABLATE & 16) eo = 0;
    const uint2 e = *(const uint2 *)(rows + eo);
    br = e.x;
    sc = __uint_as_float(e.y);
}

template <int G, int U, bool WIDE>
__device__ __forceinline__ void accumulate_chunks(u32 *S, u32 nb, const typename ItemT<WIDE>::type *items, int cnt,
                                                  u32 li, const unsigned char *rows, float QT, float T) {
    typedef typename ItemT<WIDE>::type item_t;
    const u32 li8 = li * 8;
    u32 br[U];
    float sc[U];
    item_t it[U];
#pragma unroll
    for (int u = 0; u < U; u++) chunk_issue<G, WIDE>(items[u], u < cnt, li, li8, rows, br[u], sc[u]);
#pragma unroll
    for (int u = 0; u < U; u++) it[u] = items[U + u];
    int s0 = 0;
    while (true) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            apply_slot(S, br[u], sc[u], QT, T);
            chunk_issue<G, WIDE>(it[u], s0 + U + u < cnt, li, li8, rows, br[u], sc[u]);
            it[u] = items[s0 + 2 * U + u];
        }
        s0 += U;
        if (!__any(s0 < cnt)) break;  // what is left in the ring belongs to steps >= cnt of every group: all skips
    }
}

// ------------------------------------------------------------------------------------------------
// accumulate, groups of 16 / 32 / 64 lanes with 32-bit offsets (every geometry the engine picks by itself): rows are
// stored in aligned 128-byte units of 16 entries, a chunk is G/16 units -- one 128-byte request per 16 lanes -- and its
// item is the first unit's byte offset | (units of the chunk that belong to the row - 1) in the low bits.  Rows are read
// through a raw buffer descriptor over the blob; lanes whose unit lies beyond the row, and list slots past a group's own
// count (ITEM_FILLER), get an offset outside the buffer: the load returns zeros without touching memory and the update
// goes to the scratch slot.  A step is then: item (+ unit test for G > 16) + lane offset, one buffer load, one LDS
// read-modify-write -- no bounds test, no count test, no 64-bit address arithmetic.  `wcnt` = the largest count among the
// wave's groups.
// ------------------------------------------------------------------------------------------------
constexpr u32 ITEM_FILLER = 0xFFFFFF00u;
typedef u32 v2u32 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rows_resource(const DbView &db) {
    return __builtin_amdgcn_make_buffer_rsrc((void *)db.rows, (short)0, (int)(u32)db.rows_bytes, 0x00020000);
}

template <int G, int U, bool MONO, bool WIN = false>
__device__ __forceinline__ void accumulate_units(u32 *S, const u32 *items, int wcnt, u32 li,
                                                 __amdgpu_buffer_rsrc_t rs, float QT, float T, u32 wlo4p4 = 4u, u32 w4 = 0xFFFFFFFFu) {
    const u32 li8 = li * 8;
    const u32 my_unit = li >> 4;  // which 128-byte unit of the chunk this lane reads
    u32 sb[U], it[U];
    float sc[U];
    auto issue = [&](u32 item, u32 &b, float &v) {
        u32 off = item + li8;  // G == 16: the low bits of an item are zero
        if (G > 16) off = (my_unit <= (item & 7u)) ? (item & ~127u) + li8 : ITEM_FILLER;
        if (RK_ABLATE & 16) off = li8;
        const v2u32 e = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)off, 0, RK_ROW_AUX);
        b = e.x;  // (left untouched until its step: anything computed from it here would wait for the load and serialise the ring)
        v = __uint_as_float(e.y);
    };
    auto slot_of = [&](u32 b) {  // WIN: S holds one window of the tree -- slot offsets are rebased, everything outside goes to the scratch word
        if (!WIN) return b;
        const u32 t = b - wlo4p4;
        return (t < w4) ? t + 4u : 0u;
    };
#pragma unroll
    for (int u = 0; u < U; u++) issue(items[u], sb[u], sc[u]);
#pragma unroll
    for (int u = 0; u < U; u++) it[u] = items[U + u];
    int s0 = 0;
    while (true) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            apply_slot<MONO>(S, slot_of(sb[u]), sc[u], QT, T);
            issue(it[u], sb[u], sc[u]);
            it[u] = items[s0 + 2 * U + u];
        }
        s0 += U;
        if (s0 >= wcnt) break;  // what is left in the ring are fillers
    }
}

// ------------------------------------------------------------------------------------------------
// select: top-K + LWR + output (A8/A9: PlacementProcess.java:396-451, :974-1025)
// Order among candidates is the total order of the packed key: score desc (Float.compare), then branch id asc.
// ------------------------------------------------------------------------------------------------
// ---- cross-lane helpers inside a lane group -------------------------------------------------------------------
// For G == 16 a group is exactly one DPP row: rotations are single VALU moves (no LDS crossbar traffic, no
// lgkmcnt wait).  Other group widths use ds_bpermute shuffles.
template <int S>
__device__ __forceinline__ u32 row_ror32(u32 v) {
    return (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x120 + S, 0xF, 0xF, false);  // row_ror:S
}
template <int G, int S>
__device__ __forceinline__ u64 group_rotate(u64 v, u32 li) {  // value of lane (li - S) mod G ... any fixed permutation works
    if (G == 16) {
        return ((u64)row_ror32<S>((u32)(v >> 32)) << 32) | row_ror32<S>((u32)v);
    } else {
        return shfl64(v, (int)((li + G - S) % G), G);
    }
}
// number of lanes of the group whose key is greater than mine (keys are unique or zero)
template <int G, int S>
struct RankAbove {
    static __device__ __forceinline__ int run(u64 v, u32 li) {
        return ((group_rotate<G, S>(v, li) > v) ? 1 : 0) + RankAbove<G, S - 1>::run(v, li);
    }
};
template <int G>
struct RankAbove<G, 0> {
    static __device__ __forceinline__ int run(u64, u32) { return 0; }
};
template <int G>
__device__ __forceinline__ u64 group_max(u64 v) {
    if (G == 16) {
        u64 o;
        o = ((u64)row_ror32<8>((u32)(v >> 32)) << 32) | row_ror32<8>((u32)v); v = o > v ? o : v;
        o = ((u64)row_ror32<4>((u32)(v >> 32)) << 32) | row_ror32<4>((u32)v); v = o > v ? o : v;
        o = ((u64)row_ror32<2>((u32)(v >> 32)) << 32) | row_ror32<2>((u32)v); v = o > v ? o : v;
        o = ((u64)row_ror32<1>((u32)(v >> 32)) << 32) | row_ror32<1>((u32)v); v = o > v ? o : v;
        return v;
    } else {
#pragma unroll
        for (int s = 1; s < G; s <<= 1) {
            u64 o = shfl_xor64(v, s, G);
            v = o > v ? o : v;
        }
        return v;
    }
}

// Exact top-K of list[0..c): rank r < K goes to win[r].  Returns min(c, K).  Keys are unique (they embed the
// branch id).  c <= G (the usual case): one candidate per lane, ranked with lane rotations; otherwise every lane
// ranks the candidates it owns against all others through LDS broadcast reads.
template <int G>
__device__ __forceinline__ int rank_candidates(const u64 *list, int c, u64 *win, int K, u32 li) {
    if (c <= G) {
        const u64 mine = ((int)li < c) ? list[li] : 0ull;
        const int rank = RankAbove<G, G - 1>::run(mine, li);
        if (mine != 0 && rank < K) win[rank] = mine;
    } else {
        for (int j = (int)li; j < c; j += G) {
            const u64 mine = list[j];
            int rank = 0;
            for (int t = 0; t < c; t++) rank += (list[t] > mine) ? 1 : 0;
            if (rank < K) win[rank] = mine;
        }
    }
    wave_lds_fence();
    return c < K ? c : K;
}

// Scans S[0..nb) with the G lanes of a group and resets it to UNTOUCHED; leaves the rank-r winner key in lane r.
//   pass 1  per-lane maximum, 4 scores per LDS read (float compare; the -inf UNTOUCHED marker never wins);
//   tau     the K-th largest of the G lane maxima: at least K scores are >= tau, so it bounds the answer from below;
//   pass 2  entries with key >= tau are compacted into the (now idle) hit list while S is reset, 4 per LDS access;
//           if the list fills up it is pruned to its exact top-K and tau is raised -- correct for any distribution;
//   rank    exact top-K of the few survivors.
// `list` has `cap` u64 slots; the last 16 are the winners' scratch.  S must be 16-byte aligned with s_stride % 4 == 0
// (slots in [nb, s_stride) are scratch).  Returns numBest (group-uniform).
template <int G>
__device__ __forceinline__ int select_topk_scan(u32 *S, u32 ns, u32 li, u32 gi, int K, u64 *list, int cap, u64 &win_key RK_STAMP_PARAMS) {
    // slot layout: S[0] is the scratch word (the caller has set it to UNTOUCHED), branch x is S[x + 1], ns = n_branches + 1
    const u32 nb = ns;
    u64 *win = list + (cap - 16);
    const int capc = cap - 16;
    const uint4 *S4 = (const uint4 *)S;
    uint4 *S4w = (uint4 *)S;
    const u32 n4 = (nb + 3) / 4;
    // ---- pass 1 ----
    float mo = -INFINITY;
    u32 mi = 0xFFFFFFFFu;
    for (u32 q = li; q < n4; q += G) {
        const uint4 v4 = S4[q];
        const u32 i = 4 * q;
        const float v[4] = {__uint_as_float(v4.x), __uint_as_float(v4.y), __uint_as_float(v4.z), __uint_as_float(v4.w)};
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const bool gt = (v[e] > mo) && (i + e < nb);  // false for the -inf marker; ties keep the smaller branch id
            mo = gt ? v[e] : mo;
            mi = gt ? i + e : mi;
        }
    }
    const u64 km = (mi != 0xFFFFFFFFu) ? make_key(__float_as_uint(mo), mi - 1u) : 0ull;
    // ---- tau = K-th largest lane maximum (0 if fewer than K lanes saw anything) ----
    const int rank = RankAbove<G, G - 1>::run(km, li);
    u64 tau = group_max<G>((rank == K - 1) ? km : 0ull);
    // ---- pass 2 ----
    int c = 0;
    const uint4 reset4 = make_uint4(S_UNTOUCHED, S_UNTOUCHED, S_UNTOUCHED, S_UNTOUCHED);
    for (u32 q0 = 0; q0 < n4; q0 += G) {
        const u32 q = q0 + li;
        uint4 v4 = reset4;
        if (q < n4) { v4 = S4[q]; S4w[q] = reset4; }
        const u32 i = 4 * q;
        const u32 raw[4] = {v4.x, v4.y, v4.z, v4.w};
        u64 key[4];
        bool any = false;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            key[e] = (raw[e] != S_UNTOUCHED && i + e < nb) ? make_key(raw[e], i + e - 1u) : 0ull;
            any = any || (key[e] != 0 && key[e] >= tau);
        }
        if (group_bits<G>(__ballot(any), gi) == 0) continue;  // group-uniform: no candidate in these 4G entries
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const bool cand = key[e] != 0 && key[e] >= tau;
            const u64 sub = group_bits<G>(__ballot(cand), gi);
            if (sub == 0) continue;
            if (c + G > capc) {  // prune: keep the exact top-K, raise tau
                wave_lds_fence();
                const int kept = rank_candidates<G>(list, c, win, K, li);
                if ((int)li < kept) list[li] = win[li];
                if (kept == K) tau = win[K - 1];
                wave_lds_fence();
                c = kept;
            }
            const bool still = cand && key[e] >= tau;
            const u64 sub2 = group_bits<G>(__ballot(still), gi);
            if (still) list[c + count_below<G>(sub2, li)] = key[e];
            c += __builtin_popcountll(sub2);
        }
    }
    wave_lds_fence();
    const int num = rank_candidates<G>(list, c, win, K, li);
    win_key = ((int)li < num) ? win[li] : 0ull;
    wave_lds_fence();
    return num;
}

// Conditions as wave masks (one bit per lane, in an SGPR pair) and selects driven by them: for code whose conditions feed both selects
// and wave-wide decisions (place_hash64_kernel).  All 64 lanes must be active where these are used.
__device__ __forceinline__ u64 mask_eq0(u32 v) {
    u64 m;
    asm("v_cmp_eq_u32_e64 %0, 0, %1" : "=s"(m) : "v"(v));
    return m;
}
__device__ __forceinline__ u64 mask_eq_lo16(u32 a, u32 b) {  // the low 16 bits of a and b are equal
    u64 m;
    asm("v_cmp_eq_u16_e64 %0, %1, %2" : "=s"(m) : "v"(a), "v"(b));
    return m;
}
__device__ __forceinline__ u64 mask_ge(u32 a, u32 b) {  // a >= b (unsigned), b wave-uniform
    u64 m;
    asm("v_cmp_ge_u32_e64 %0, %1, %2" : "=s"(m) : "v"(a), "s"(b));
    return m;
}
__device__ __forceinline__ u32 mask_select(u32 if_clear, u32 if_set, u64 m) {
    u32 r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(if_clear), "v"(if_set), "s"(m));
    return r;
}
// A returning LDS compare-and-swap whose result the CALLER waits for (lds_cas_wait over all results before their first use), so that
// several are in flight at once.  at = byte offset in the workgroup's LDS (kernels without static LDS: the dynamic block starts at 0).
__device__ __forceinline__ u32 lds_cas_issue(u32 at, u32 expect, u32 put) {
    u32 r;
    asm volatile("ds_cmpst_rtn_b32 %0, %1, %2, %3" : "=v"(r) : "v"(at), "v"(expect), "v"(put) : "memory");
    return r;
}
template <int N>
__device__ __forceinline__ void lds_cas_wait(u32 (&r)[N]) {
    asm volatile("s_waitcnt lgkmcnt(0)" : : : "memory");
#pragma unroll
    for (int i = 0; i < N; i++) asm volatile("" : "+v"(r[i]));  // (the results are defined from here on)
}
__device__ __forceinline__ u32 wave_max_u32(u32 v) {  // wave-uniform maximum: four DPP row steps, then the four rows' results through SGPRs
    v = max(v, row_ror32<8>(v));
    v = max(v, row_ror32<4>(v));
    v = max(v, row_ror32<2>(v));
    v = max(v, row_ror32<1>(v));
    const u32 m0 = (u32)__builtin_amdgcn_readlane((int)v, 0), m1 = (u32)__builtin_amdgcn_readlane((int)v, 16);
    const u32 m2 = (u32)__builtin_amdgcn_readlane((int)v, 32), m3 = (u32)__builtin_amdgcn_readlane((int)v, 48);
    return max(max(m0, m1), max(m2, m3));
}

// group-wide maximum of a 32-bit value.  G == 16: a group is one DPP row, four rotate-and-max steps, no LDS traffic; G == 64: the same per row, then the four rows through SGPRs.
template <int G>
__device__ __forceinline__ u32 group_max_u32(u32 v) {
    if (G == 16) {
        v = max(v, row_ror32<8>(v));
        v = max(v, row_ror32<4>(v));
        v = max(v, row_ror32<2>(v));
        v = max(v, row_ror32<1>(v));
    } else if (G == 64) {
        v = wave_max_u32(v);
    } else {
#pragma unroll
        for (int s = 1; s < G; s <<= 1) v = max(v, (u32)__shfl_xor((int)v, s, G));
    }
    return v;
}

// Fast select: ONE pass over S.  A lane reads S four consecutive words at a time; word c of every quad feeds the lane's STREAM c,
// which keeps its two best (score, quad) pairs in registers plus the best score it had to drop.  Slot 4*quad + c belongs to
// stream c of lane quad % G, so neighbouring branches -- the usual shape of a placement: a clade -- land in different streams.
// K rounds of group-max over the 4 * G stream heads then give the exact top-K unless some stream might still hide a better
// entry (it dropped something >= the K-th winner: the stream would need >= 3 of the K best); then the caller runs the exact
// two-pass scan (select_topk_scan) or its own exact path.  A round is two 32-bit reductions: the largest head score, then
// among the streams holding it the smallest slot (= the order of the packed key: score desc, branch asc).
struct Heads4 {
    float s0[4], s1[4], dr[4];  // per stream: the two best scores (s0 >= s1) and the best dropped one
    u32 q0[4], q1[4];           // their quads
};
__device__ __forceinline__ void heads_clear(Heads4 &h) {
#pragma unroll
    for (int c = 0; c < 4; c++) {
        h.s0[c] = h.s1[c] = h.dr[c] = -INFINITY;
        h.q0[c] = h.q1[c] = 0x3FFFFFFFu;
    }
}
// sorted insertion; strict '>' keeps the smaller slot ahead among equal scores and is false for the -inf marker of untouched
// branches (one select per statement: nested conditionals came out of hipcc as divergent branches)
template <int C>
__device__ __forceinline__ void heads_feed(Heads4 &h, float v, u32 q) {
    const bool g0 = v > h.s0[C], g1 = v > h.s1[C];
    const u32 t1 = g1 ? q : h.q1[C];
    h.q1[C] = g0 ? h.q0[C] : t1;
    h.q0[C] = g0 ? q : h.q0[C];
    h.dr[C] = __builtin_amdgcn_fmed3f(v, h.s1[C], h.dr[C]);
    h.s1[C] = __builtin_amdgcn_fmed3f(v, h.s0[C], h.s1[C]);
    h.s0[C] = g0 ? v : h.s0[C];
}
__device__ __forceinline__ void heads_feed_quad(Heads4 &h, const uint4 v4, u32 q) {
    heads_feed<0>(h, __uint_as_float(v4.x), q);
    heads_feed<1>(h, __uint_as_float(v4.y), q);
    heads_feed<2>(h, __uint_as_float(v4.z), q);
    heads_feed<3>(h, __uint_as_float(v4.w), q);
}
// scans the ns words of S (word 0 = scratch, word i = branch slot_base + i - 1 of the tree; slot_base a multiple of 4) into the
// heads; the quads are numbered slot_base / 4 + i so that a head's slot is 4 * quad + stream over the whole tree
// RESET: every quad is set back to UNTOUCHED right behind its read (for callers that never need S again)
template <int G, bool RESET = false>
__device__ __forceinline__ void heads_scan(u32 *S, u32 ns, u32 li, u32 slot_base, Heads4 &h) {
    uint4 *S4 = (uint4 *)S;
    const uint4 reset4 = make_uint4(S_UNTOUCHED, S_UNTOUCHED, S_UNTOUCHED, S_UNTOUCHED);
    const u32 n4_full = ns / 4;  // quads that lie entirely below ns need no bounds test
    const u32 qb = slot_base / 4;
    if (li < n4_full) {
        uint4 cur = S4[li];
        for (u32 q = li; q < n4_full; q += G) {
            const uint4 v4 = cur;
            if (q + G < n4_full) cur = S4[q + G];  // next quad is in flight while this one is ranked
            if (RESET) S4[q] = reset4;
            heads_feed_quad(h, v4, qb + q);
        }
    }
    if ((ns & 3u) && (n4_full % G) == li) {  // the partial last quad
        uint4 v4 = S4[n4_full];
        if (RESET) S4[n4_full] = reset4;
        const u32 i = 4 * n4_full;
        if (i >= ns) v4.x = S_UNTOUCHED;
        if (i + 1 >= ns) v4.y = S_UNTOUCHED;
        if (i + 2 >= ns) v4.z = S_UNTOUCHED;
        if (i + 3 >= ns) v4.w = S_UNTOUCHED;
        heads_feed_quad(h, v4, qb + n4_full);
    }
}
// K rounds over the 4 * G stream heads; returns numBest, the rank-r key in lane r (0 beyond numBest), and whether a dropped
// entry could belong to the answer: K ranks filled -- only if it ties or beats the weakest winner; fewer -- any dropped entry.
// The rounds run on the order-preserving integer image of the scores (integer max folds into the DPP rotate).
template <int G, bool FULLQ = false>  // FULLQ: a head's q is the slot itself (streams fed in any order), not its quad
__device__ __forceinline__ int heads_rounds_raw(const Heads4 &h, int K, u32 li, u32 gi, u32 &win_o, u32 &win_i, bool &doubt, u32 *kth_o = nullptr) {
    constexpr u32 ORD_NEG_INF = 0x007FFFFFu;  // ord_f32(-inf)
    u32 o0[4], o1[4], n0[4], n1[4];  // n = ~slot: the smaller slot wins a max
#pragma unroll
    for (int c = 0; c < 4; c++) {
        o0[c] = ord_f32(h.s0[c]); o1[c] = ord_f32(h.s1[c]);
        n0[c] = FULLQ ? ~h.q0[c] : ~(4u * h.q0[c] + (u32)c);
        n1[c] = FULLQ ? ~h.q1[c] : ~(4u * h.q1[c] + (u32)c);
    }
    u32 last = ORD_NEG_INF;
    int num = 0;
    win_o = 0; win_i = 0;  // lane r: ordered score and slot (4 * quad + stream) of rank r
    for (int r = 0; r < K; r++) {
        const u32 m = group_max_u32<G>(max(max(o0[0], o0[1]), max(o0[2], o0[3])));
        const bool valid = m != ORD_NEG_INF;  // group-uniform: something is left
        u32 cand = 0;
#pragma unroll
        for (int c = 0; c < 4; c++) cand = max(cand, (o0[c] == m) ? n0[c] : 0u);
        const u32 w = group_max_u32<G>(valid ? cand : 0u);  // smallest slot among the streams holding m
        const bool mine = (int)li == r && valid;
        win_o = mine ? m : win_o;
        win_i = mine ? ~w : win_i;
        num += valid ? 1 : 0;
        last = valid ? m : last;
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const bool pop = valid && o0[c] == m && n0[c] == w;  // exactly one stream of one lane
            o0[c] = pop ? o1[c] : o0[c]; n0[c] = pop ? n1[c] : n0[c];
            o1[c] = pop ? ORD_NEG_INF : o1[c];
        }
    }
    const u32 od = max(max(ord_f32(h.dr[0]), ord_f32(h.dr[1])), max(ord_f32(h.dr[2]), ord_f32(h.dr[3])));
    const bool d = (num == K) ? (od >= last) : (od != ORD_NEG_INF);
    doubt = group_bits<G>(__ballot(d), gi) != 0;  // group-uniform
    if (kth_o) *kth_o = (num == K) ? last : ORD_NEG_INF;  // ordered score of the K-th winner (anything counts while fewer than K are found)
    return num;
}
// the same for the slot layout of the packed / ASCII kernels (word 0 of S is the scratch slot, branch x lives in word x + 1)
template <int G>
__device__ __forceinline__ int heads_rounds(const Heads4 &h, int K, u32 li, u32 gi, u64 &win_key, bool &doubt) {
    u32 win_o, win_i;
    const int num = heads_rounds_raw<G>(h, K, li, gi, win_o, win_i, doubt);
    win_key = ((int)li < num) ? (((u64)win_o << 32) | (u64)(0xFFFFu - (win_i - 1u))) : 0ull;
    return num;
}
template <int G>
__device__ __forceinline__ int select_topk(u32 *S, u32 n_branches, u32 li, u32 gi, int K, u64 *list, int cap, u64 &win_key RK_STAMP_PARAMS) {
    // slot layout: the scratch word S[0] leaves the competition, then the scan runs over ns = n_branches + 1 slots and
    // slot i stands for branch i - 1
    if (li == 0) S[0] = S_UNTOUCHED;
    wave_lds_fence();
    const u32 nb = n_branches + 1;
    if (K > RK_HEADS_MAX_K || K > G) return select_topk_scan<G>(S, nb, li, gi, K, list, cap, win_key RK_STAMP_ARGS);
    Heads4 h;
    heads_clear(h);
    heads_scan<G>(S, nb, li, 0u, h);
    RK_STAMP(8);
    bool doubt;
    const int num = heads_rounds<G>(h, K, li, gi, win_key, doubt);
    if (doubt) return select_topk_scan<G>(S, nb, li, gi, K, list, cap, win_key RK_STAMP_ARGS);
    RK_STAMP(9);
    uint4 *S4w = (uint4 *)S;
    const u32 n4 = (nb + 3) / 4;
    const uint4 reset4 = make_uint4(S_UNTOUCHED, S_UNTOUCHED, S_UNTOUCHED, S_UNTOUCHED);
    for (u32 q = li; q < n4; q += G) S4w[q] = reset4;
    return num;
}

// ---- the fast select cut in two, for kernels that see the score vector one window at a time (place_packed16w_kernel):
//      per window the scan feeds the heads and resets S; the K rounds run once at the end ----
template <int G>
__device__ __forceinline__ void heads_scan_reset(u32 *S, u32 n_branches, u32 li, u32 slot_base, Heads4 &h) {
    if (li == 0) S[0] = S_UNTOUCHED;
    wave_lds_fence();
    heads_scan<G, true>(S, n_branches + 1, li, slot_base, h);  // (a tile in doubt is accumulated again from its lists, not from S)
}

template <int LANE>
__device__ __forceinline__ u32 row_bcast32(u32 v) {  // value of lane LANE of the caller's 16-lane row
    return (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x150 + LANE, 0xF, 0xF, false);  // row_newbcast:LANE
}

// LWR + keep-factor + output rows for one read.  Called by all lanes of the group (li = lane in group).
// Requires G >= keep_at_most.
template <int G>
__device__ __forceinline__ void weigh_and_store(const PlaceArgs &a, u64 r, u32 li, int numBest, u64 win_key,
                                                u32 flags) {
    const int K = (int)a.keep_at_most;
    float score = unord_f32((u32)(win_key >> 32));
    u32 branch = 0xFFFFu - (u32)(win_key & 0xFFFFu);
    int n_rows = 0;
    double lwr = 0.0;
    if (numBest > 0 && !(RK_ABLATE & 4)) {
        flags |= RK_FLAG_PLACED;
        const bool mine = (int)li < numBest;
        float best, lowest;
        if (G == 16) {  // a group is one DPP row: no LDS round trips (the shuffles below are ds_bpermute + a wait each)
            best = __uint_as_float(row_bcast32<0>(__float_as_uint(score)));
            float lo = mine ? score : INFINITY;  // (the winners are in descending order: the last one is the smallest)
            lo = fminf(lo, __uint_as_float(row_ror32<8>(__float_as_uint(lo))));
            lo = fminf(lo, __uint_as_float(row_ror32<4>(__float_as_uint(lo))));
            lo = fminf(lo, __uint_as_float(row_ror32<2>(__float_as_uint(lo))));
            lowest = fminf(lo, __uint_as_float(row_ror32<1>(__float_as_uint(lo))));
        } else {
            best = __shfl(score, 0, G);
            lowest = __shfl(score, numBest - 1, G);
        }
        // computeWeightRatioShift (PlacementProcess.java:384-390); `lowest` starts at 0.0f in :413 and every
        // score is < 0 whenever the shift matters, so min(0, lowest) <= -308 <=> lowest <= -308.
        float lowest0 = lowest < 0.0f ? lowest : 0.0f;
        float shift = (-308.0f >= lowest0) ? best : 0.0f;
        // :441-448  sum_{ascending} pow(10, (double)(float)(score - shift));  (:418 sums in heap order when
        // shift == 0 -- same terms, order differs only in the last ulp of a double)
        float d32 = score - shift;
        double term = mine ? exp10((double)d32) : 0.0;
        double sum = 0.0;
        if (G == 16) {
            // the same additions in the same order as the loop of the other branch -- t[n-1] + t[n-2] + ... + t[0], the terms beyond
            // the winners being 0.0 -- as a chain over the row: acc(i) = t(i) + acc(i + 1), K - 1 times; lane 0 ends with the sum
            auto shl1 = [](double v) {  // value of the next lane of the row (0.0 behind its last lane)
                const u64 b = (u64)__double_as_longlong(v);
                const u32 lo = (u32)__builtin_amdgcn_update_dpp(0, (int)(u32)b, 0x101, 0xF, 0xF, true);
                const u32 hi = (u32)__builtin_amdgcn_update_dpp(0, (int)(u32)(b >> 32), 0x101, 0xF, 0xF, true);
                return __longlong_as_double((long long)(((u64)hi << 32) | lo));
            };
            double acc = term;
            for (int s_ = 1; s_ < K; s_++) acc = term + shl1(acc);
            const u64 ab = (u64)__double_as_longlong(acc);
            sum = __longlong_as_double((long long)(((u64)row_bcast32<0>((u32)(ab >> 32)) << 32) | row_bcast32<0>((u32)ab)));
        } else {
            for (int q = numBest - 1; q >= 0; q--) sum += shfl_f64(term, q, G);
        }
        float shift2 = (-308.0f >= lowest) ? best : 0.0f;  // :978-980 uses the true minimum
        // :392-394 Math.pow(10.0, (double)score - (double)shift2): the same number as `term` unless a shift is in play
        double numer = term;
        if (__any(shift != 0.0f || shift2 != 0.0f)) numer = mine ? exp10((double)score - (double)shift2) : 0.0;
        double ratio = mine ? numer / sum : 0.0;
        double best_ratio;
        if (G == 16) {
            const u64 rb = (u64)__double_as_longlong(ratio);
            best_ratio = __longlong_as_double((long long)(((u64)row_bcast32<0>((u32)(rb >> 32)) << 32) | row_bcast32<0>((u32)rb)));
        } else {
            best_ratio = shfl_f64(ratio, 0, G);
        }
        bool fail = mine && li > 0 && (ratio < best_ratio * (double)a.keep_factor);  // :998-1000
        const u32 lane = threadIdx.x & 63;
        // (the failing ranks sit below numBest <= 16: 32 bits of the group's part of the ballot are all there is to look at)
        const u32 failm = (u32)group_bits<G>(__ballot(fail), (lane - li) / (u32)G) | (1u << numBest);
        n_rows = __builtin_ctz(failm);
        lwr = ratio;
        if (!(best >= a.ns_bound)) {  // :974
            n_rows = 0;
            flags |= RK_FLAG_BELOW_NSBOUND;
        }
    }
    if (RK_ABLATE & 8192) {  // timing only: everything but the stores
        asm volatile("" : : "v"(branch), "v"(score), "v"(lwr), "v"(n_rows), "v"(flags));
        return;
    }
    if ((int)li < K) {
        bool on = (int)li < n_rows;
        a.o_branch[r * K + li] = on ? (unsigned short)branch : (unsigned short)0xFFFFu;
        a.o_score[r * K + li] = on ? score : -INFINITY;
        a.o_lwr[r * K + li] = on ? lwr : 0.0;
    }
    if (li == 0) {
        a.o_nrows[r] = (unsigned char)n_rows;
        a.o_flags[r] = flags;
    }
}

// ------------------------------------------------------------------------------------------------
// main placement kernel: reads packed 2-bit / 5-bit, no ambiguity characters
// ------------------------------------------------------------------------------------------------
template <int G, int BITS, int TM, bool WIDE, int U, int PU>
__global__ void __launch_bounds__(256) place_packed_kernel(PlaceArgs a) {
    constexpr int NG = 64 / G;
    constexpr int LOG2G = G == 8 ? 3 : (G == 16 ? 4 : (G == 32 ? 5 : 6));
    typedef typename ItemT<WIDE>::type item_t;
    extern __shared__ u32 lds[];
    const bool perm_given = tile_order_given(a);
    const u32 lane = threadIdx.x & 63;
    const u32 wave = threadIdx.x >> 6;
    const u32 waves_per_block = blockDim.x >> 6;
    const u32 gi = lane / G, li = lane % G;
    // per wave: NG score vectors | NG hit lists (list_cap u64 slots each)
    const u32 wave_words = NG * (a.s_stride + 2 * a.list_cap);
    u32 *wbase = lds + wave * wave_words;
    u32 *S = wbase + gi * a.s_stride;
    u64 *list = (u64 *)(wbase + NG * a.s_stride) + gi * a.list_cap;
    item_t *items = (item_t *)list;
    const u32 nb = a.db.n_branches;
    const u32 k = a.db.k;
    const float T = a.db.T;
    constexpr bool FAST16 = (G >= 16 && !WIDE);  // buffer-addressed unit chunks (accumulate_units)
    const __amdgpu_buffer_rsrc_t rows_rs = rows_resource(a.db);
    // chunk items the list can take (3U slots of slack for the read-ahead of accumulate_chunks)
    const int cap_items = (int)(a.list_cap * (sizeof(u64) / sizeof(item_t))) - 3 * U - 2;
    const int cap_rows = (int)a.list_cap - 1;  // row descriptors (fallback path), one slot for the sentinel

    for (u32 i = li; i < a.s_stride; i += G) S[i] = S_UNTOUCHED;
    wave_lds_fence();

    const u64 n_tiles = (a.n_reads + NG - 1) / NG;
    const u64 wave_global = (u64)blockIdx.x * waves_per_block + wave;
    const u64 wave_count = (u64)gridDim.x * waves_per_block;
#ifdef RK_STAMPS
    unsigned long long st_[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long t_ = rk_now();
#endif
    for (u64 tile = wave_global; tile < n_tiles; tile += wave_count) {
        const bool have = tile * NG + gi < a.n_reads;
        const u64 r = have ? tile_read(a, tile * NG + gi, perm_given) : 0ull;
        u32 R = 0, fin = 0;
        if (have) {
            R = a.lens ? a.lens[r] : a.fixed_len;
            fin = a.flags_in ? a.flags_in[r] : 0u;
            const u32 cap_syms = (a.words_per_read * 32u) / BITS;  // never read past the packed record
            R = R < cap_syms ? R : cap_syms;
        }
        u32 flags = fin & (RK_FLAG_BAD_CHAR | RK_FLAG_AMBIGUOUS | RK_FLAG_TOO_LONG);
        const bool is_amb = (fin & RK_FLAG_AMBIGUOUS) != 0;
        const bool rejected = (fin & (RK_FLAG_BAD_CHAR | RK_FLAG_TOO_LONG)) != 0;
        if (R < k) flags |= RK_FLAG_TOO_SHORT;
        // Q = sk.getMerCount() (AmbigSequenceKnife.java:191)
        const u32 Q = (have && !is_amb && !rejected && R >= k) ? (R - k + 1) : 0u;
        const float QT = (float)(int)Q * T;  // int * float (PlacementProcess.java:728)
        const u32 *rec = a.packed + (have ? r : 0ull) * a.words_per_read;  // in-bounds for idle groups too
        // warm the cache with the NEXT tile's packed records (streamed from HBM, ~2 us if met cold by the probe)
        u32 warm = 0;
        {
            const u64 sn = (tile + wave_count) * NG + gi;
            if (sn < a.n_reads) warm = a.packed[tile_read(a, sn, perm_given) * a.words_per_read + (li < a.words_per_read ? li : 0u)];
        }

        u32 pos = 0;
        int cnt = 0;  // chunk items waiting in the list
        auto flush = [&]() {
            if (FAST16) {
                // the longest list of the wave's four groups sets the step count; shorter lists are padded with fillers
                int wcnt = __builtin_amdgcn_readlane(cnt, 0);
                if (G <= 32) wcnt = max(wcnt, __builtin_amdgcn_readlane(cnt, 32));
                if (G == 16) {
                    wcnt = max(wcnt, __builtin_amdgcn_readlane(cnt, 16));
                    wcnt = max(wcnt, __builtin_amdgcn_readlane(cnt, 48));
                }
                for (int i = cnt + (int)li; i < wcnt + 2 * U; i += G) ((u32 *)items)[i] = ITEM_FILLER;
                wave_lds_fence();
                RK_STAMP(3);
                if (!(RK_ABLATE & 1)) {
                    if (a.db.mono) accumulate_units<G, U, true>(S, (const u32 *)items, wcnt, li, rows_rs, QT, T);
                    else accumulate_units<G, U, false>(S, (const u32 *)items, wcnt, li, rows_rs, QT, T);
                }
                wave_lds_fence();
                RK_STAMP(4);
                cnt = 0;
                return;
            }
            wave_lds_fence();
            RK_STAMP(3);
            if (!(RK_ABLATE & 1)) accumulate_chunks<G, U, WIDE>(S, nb, items, cnt, li, a.db.rows, QT, T);
            wave_lds_fence();
            RK_STAMP(4);
            cnt = 0;
        };
        RK_STAMP(0);  // tile setup
        while (true) {
            const bool more = pos < Q;
            if (!__any(more)) break;
            // ---- probe PU*G positions: every descriptor gather of the batch is in flight together ----
            u64 desc[PU];
#pragma unroll
            for (int u = 0; u < PU; u++) {  // out-of-range lanes re-read position 0 and drop the result later
                const u32 j = pos + u * G + li;
                desc[u] = extract_code<BITS>(rec, a.words_per_read, (more && j < Q) ? j : 0u, k);
            }
            if (TM != TM_HASH) {
                RawSlot raw[PU];
#pragma unroll
                for (int u = 0; u < PU; u++) raw[u] = lookup_fetch<BITS, TM>(a.db, desc[u]);
                __builtin_amdgcn_sched_barrier(0);  // all PU gathers are issued before any of them is decoded
#pragma unroll
                for (int u = 0; u < PU; u++) {
                    const u32 j = pos + u * G + li;
                    const u64 d = lookup_decode<BITS, TM>(a.db, raw[u], desc[u]);
                    desc[u] = (more && j < Q) ? d : 0ull;
                }
            } else {
                // open addressing: the home slots of the whole batch are gathered together (at load <= 0.5 most
                // probes end there); lanes whose slot holds another key walk on, the wave loops until all are done
                uint4 slot[PU];
                u64 h[PU];
#pragma unroll
                for (int u = 0; u < PU; u++) {
                    h[u] = mix64(desc[u]) & a.db.hash_mask;
                    slot[u] = a.db.slots[h[u]];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < PU; u++) {
                    const u32 j = pos + u * G + li;
                    const u64 want = desc[u] + 1;
                    u64 key = ((u64)slot[u].y << 32) | slot[u].x;
                    u64 d = ((u64)slot[u].w << 32) | slot[u].z;
                    bool walking = (more && j < Q) && key != want && key != 0;
                    while (__any(walking)) {
                        if (walking) {
                            h[u] = (h[u] + 1) & a.db.hash_mask;
                            const uint4 s2 = a.db.slots[h[u]];
                            key = ((u64)s2.y << 32) | s2.x;
                            d = ((u64)s2.w << 32) | s2.z;
                            walking = key != want && key != 0;
                        }
                    }
                    desc[u] = ((more && j < Q) && key == want) ? d : 0ull;
                }
            }
            asm volatile("" ::"v"((u32)desc[0]), "v"((u32)desc[PU - 1]));
            RK_STAMP(1);  // probe: codes + descriptor gathers (includes their latency)
            // ---- chunks per row and their exclusive prefix sums in k-mer order (lane order inside a sub-batch) ----
            u32 nch[PU], excl[PU];
            int total = 0;
#pragma unroll
            for (int u = 0; u < PU; u++) {
                const u32 lenp = (u32)desc[u] & DESC_LEN_MASK;  // padded row length, 0 = miss
                nch[u] = (lenp + G - 1) >> LOG2G;
                u32 incl = nch[u];
                if (G == 16) {  // a group is one DPP row: row_shr with bound_ctrl shifts zeros in, no LDS crossbar
                    incl += (u32)__builtin_amdgcn_update_dpp(0, (int)incl, 0x111, 0xF, 0xF, true);
                    incl += (u32)__builtin_amdgcn_update_dpp(0, (int)incl, 0x112, 0xF, 0xF, true);
                    incl += (u32)__builtin_amdgcn_update_dpp(0, (int)incl, 0x114, 0xF, 0xF, true);
                    incl += (u32)__builtin_amdgcn_update_dpp(0, (int)incl, 0x118, 0xF, 0xF, true);
                } else {
#pragma unroll
                    for (int sft = 1; sft < G; sft <<= 1) {
                        const u32 t = __shfl_up(incl, sft, G);
                        incl += (li >= (u32)sft) ? t : 0u;
                    }
                }
                excl[u] = (u32)total + incl - nch[u];
                total += (int)__shfl(incl, G - 1, G);
            }
            // ---- room in the list? (decisions are per group, actions are taken by the whole wave) ----
            if (__any(more && cnt + total > cap_items)) flush();
            if (__any(more && total > cap_items)) {
                // a batch that does not fit an empty list (very long rows): one descriptor per row and the row cursor,
                // in parts of at most cap_rows rows
                const int per_part = cap_rows / G;  // sub-batches per part (>= 1: list_cap > G)
                for (int u_lo = 0; u_lo < PU; u_lo += per_part) {
                    int rc = 0;
#pragma unroll
                    for (int u = 0; u < PU; u++) {
                        const bool part = u >= u_lo && u < u_lo + per_part;
                        const bool hit = part && ((u32)desc[u] & DESC_LEN_MASK) != 0;
                        const u64 bal = __ballot(hit);
                        const u64 sub = group_bits<G>(bal, gi);
                        if (hit) list[rc + count_below<G>(sub, li)] = desc[u];
                        rc += __builtin_popcountll(sub);
                    }
                    if (li == 0) list[rc] = 0;  // sentinel: an empty row ends the cursor
                    wave_lds_fence();
                    if (!(RK_ABLATE & 1) && __any(rc > 0)) accumulate_list<G, U, WIDE>(S, nb, list, rc, li, a.db.rows, QT, T);
                    wave_lds_fence();
                }
            } else {
#pragma unroll
                for (int u = 0; u < PU; u++) {
                    const u64 off8 = desc[u] >> DESC_LEN_BITS;
                    const u32 lenp = (u32)desc[u] & DESC_LEN_MASK;
                    const int base = cnt + (int)excl[u];
                    if (FAST16) {
                        const u32 rb = (u32)off8 * 8u;  // byte offset of the row (128-byte aligned units)
                        constexpr u32 UPC = G / 16;     // units per chunk
                        const u32 units = lenp >> 4;    // units of the row
                        u32 *it32 = (u32 *)items;
                        // (RK_ABLATE & 32, timing only: later chunks re-read the row's first unit -- same steps, fewer requests)
                        auto item_of = [&](u32 c) {
                            const u32 left = units - c * UPC;
                            const u32 first = ((RK_ABLATE & 32) && c) ? rb : rb + c * (UPC * 128u);
                            if (UPC == 1) return first;  // one unit per chunk: nothing to encode
                            return first | ((left < UPC ? left : UPC) - 1u);
                        };
                        // most rows are 1-2 chunks: those are written without a wave vote, the rest in a voted loop
                        if (nch[u] > 0) it32[base] = item_of(0);
                        if (nch[u] > 1) it32[base + 1] = item_of(1);
                        for (u32 c = 2; __any(c < nch[u]); c++)
                            if (c < nch[u]) it32[base + (int)c] = item_of(c);
                        continue;
                    }
                    // most rows are 1-2 chunks: those are written without a wave vote, the rest in a voted loop
                    if (nch[u] > 0) items[base] = make_item<G, WIDE>(off8, lenp < (u32)G ? lenp : (u32)G);
                    if (nch[u] > 1) items[base + 1] = make_item<G, WIDE>(off8 + G, lenp - G < (u32)G ? lenp - G : (u32)G);
                    for (u32 c = 2; __any(c < nch[u]); c++) {
                        if (c < nch[u]) {
                            const u32 left = lenp - c * G;
                            items[base + (int)c] = make_item<G, WIDE>(off8 + (u64)c * G, left < (u32)G ? left : (u32)G);
                        }
                    }
                }
                cnt += total;
            }
            if (more) pos += PU * G;
            RK_STAMP(2);  // scans + item emission
        }
        if (__any(cnt > 0)) flush();

        // ---- select + weigh + store (also resets S) ----
        u64 win_key;
        int numBest = 0;
        if (RK_ABLATE & 2) { win_key = list[0]; for (u32 i = li; i <= nb; i += G) S[i] = S_UNTOUCHED; }
        else numBest = select_topk<G>(S, nb, li, gi, (int)a.keep_at_most, list, (int)a.list_cap, win_key RK_STAMP_ARGS);
        wave_lds_fence();
        RK_STAMP(5);  // select
        const bool deferred = is_amb && a.has_ascii && !rejected;  // the ASCII kernel writes these
        if (have && !deferred) weigh_and_store<G>(a, r, li, numBest, win_key, flags);
        asm volatile("" ::"v"(warm));  // keep the warming load alive
        RK_STAMP(6);  // weigh + store
    }
#ifdef RK_STAMPS
    if (lane == 0 && wave_global < 4096)
        for (int i = 0; i < 16; i++) rk_stamp_buf[wave_global * 16 + i] = st_[i];
#endif
}

// ------------------------------------------------------------------------------------------------
// place_packed16_kernel: the same algorithm as place_packed_kernel<16, ..., WIDE=false> for the geometry every BASELINE
// small-tree config runs in (16 lanes per read, direct table, packed record of <= 16 words: 256 bases / 102 residues),
// software-pipelined ACROSS tiles:
//   * lane li of a group holds word li of its read's packed record -- ONE coalesced load per tile instead of two loads
//     per k-mer position; DNA codes of the first 144 positions come out of it with DPP row broadcasts + v_alignbit;
//   * the record of tile t+1 is loaded before the accumulate phase of tile t, its table gathers are issued before the
//     select phase of tile t and decoded after it: the probe latency of a wave hides under its own select ALU time
//     instead of relying on another wave's row gathers to fill it.
// Results are identical to place_packed_kernel's (same emit order, same accumulate, same select).
// ------------------------------------------------------------------------------------------------
template <int PU, int U0>
struct Batch0Words {
    static __device__ __forceinline__ void run(u32 recw, u32 (&w)[PU + 1]) {
        w[U0] = row_bcast32<U0>(recw);
        Batch0Words<PU, U0 + 1>::run(recw, w);
    }
};
template <int PU>
struct Batch0Words<PU, PU> {
    static __device__ __forceinline__ void run(u32 recw, u32 (&w)[PU + 1]) { w[PU] = row_bcast32<(PU < 16 ? PU : 15)>(recw); }
};

// k-mer codes of positions u*16 + li (u < PU) of a record held one word per lane (lanes >= words_per_read hold 0)
template <int BITS, int PU>
__device__ __forceinline__ void record_codes(u32 recw, u32 pos, u32 li, u32 k, u32 Q, u64 (&code)[PU]) {
    if (BITS == 2 && PU <= 15) {
        if (pos == 0) {  // (the usual batch: position u*16 + li starts in word u at bit 2*li)
            u32 w[PU + 1];
            Batch0Words<PU, 0>::run(recw, w);
            const u32 mask = (k >= 16) ? 0xFFFFFFFFu : ((1u << (2 * k)) - 1u);
#pragma unroll
            for (int u = 0; u < PU; u++) code[u] = __builtin_amdgcn_alignbit(w[u + 1], w[u], 2 * li) & mask;  // any bit pattern is a valid DNA code
            return;
        }
    }
#pragma unroll
    for (int u = 0; u < PU; u++) {
        const u32 j0 = pos + u * 16 + li;
        const u32 j = j0 < Q ? j0 : 0u;  // (AA: digits read past the read's end could be >= 20)
        const u32 bit = j * BITS, wi = bit >> 5, sh = bit & 31;
        const u32 w0 = __shfl(recw, (int)(wi & 15u), 16);
        u32 w1 = __shfl(recw, (int)((wi + 1) & 15u), 16);
        w1 = wi + 1 < 16 ? w1 : 0u;
        const u32 nbits = k * BITS;
        if (nbits <= 32) {  // (uniform) the code fits one word: a funnel shift over two record words, no third word, no 64-bit shifts
            const u32 c32 = __builtin_amdgcn_alignbit(w1, w0, sh);
            code[u] = nbits == 32 ? c32 : (c32 & ((1u << nbits) - 1u));
            continue;
        }
        u32 w2 = 0u;
        if (BITS * 12 + 31 > 64) {
            w2 = __shfl(recw, (int)((wi + 2) & 15u), 16);
            w2 = (wi + 2 < 16 && nbits + sh > 64) ? w2 : 0u;
        }
        const u64 c = funnel96(w0, w1, w2, sh);
        code[u] = nbits >= 64 ? c : (c & ((1ull << nbits) - 1));
    }
}

template <int BITS, int TM, int U, int PU>
__global__ void __launch_bounds__(256) place_packed16_kernel(PlaceArgs a) {
    constexpr int G = 16, NG = 4;
    static_assert(TM != TM_HASH, "direct tables only");
    extern __shared__ u32 lds[];
    const bool perm_given = tile_order_given(a);
    const u32 lane = threadIdx.x & 63;
    const u32 wave = threadIdx.x >> 6;
    const u32 waves_per_block = blockDim.x >> 6;
    const u32 gi = lane / G, li = lane % G;
    const u32 wave_words = NG * (a.s_stride + 2 * a.list_cap);
    u32 *wbase = lds + wave * wave_words;
    u32 *S = wbase + gi * a.s_stride;
    u64 *list = (u64 *)(wbase + NG * a.s_stride) + gi * a.list_cap;
    u32 *items = (u32 *)list;
    const u32 nb = a.db.n_branches;
    const u32 k = a.db.k;
    const float T = a.db.T;
    const __amdgpu_buffer_rsrc_t rows_rs = rows_resource(a.db);
    const int cap_items = (int)(a.list_cap * 2) - 3 * U - 2;
    const int cap_rows = (int)a.list_cap - 1;
    const u32 wpr = a.words_per_read;  // <= 16 (checked by the host)

    for (u32 i = li; i < a.s_stride; i += G) S[i] = S_UNTOUCHED;
    wave_lds_fence();

    const u64 n_tiles = (a.n_reads + NG - 1) / NG;
    const u64 wave_global = (u64)blockIdx.x * waves_per_block + wave;
    const u64 wave_count = (u64)gridDim.x * waves_per_block;

    // per-tile inputs of a group: its record word, length, incoming flags
    auto load_tile = [&](u64 tile, u32 &recw, u32 &R, u32 &fin, bool &have) {
        const u64 slot = tile * NG + gi;
        have = tile < n_tiles && slot < a.n_reads;
        const u64 r = have ? tile_read(a, slot, perm_given) : 0ull;
        recw = 0; R = 0; fin = 0;
        if (have) {
            if (li < wpr) recw = a.packed[r * wpr + li];
            R = a.lens ? a.lens[r] : a.fixed_len;
            fin = a.flags_in ? a.flags_in[r] : 0u;
        }
    };
    auto mer_count = [&](u32 R, u32 fin, bool have) -> u32 {  // Q = sk.getMerCount() (AmbigSequenceKnife.java:191)
        const u32 cap_syms = (wpr * 32u) / BITS;
        R = R < cap_syms ? R : cap_syms;
        const bool is_amb = (fin & RK_FLAG_AMBIGUOUS) != 0;
        const bool rejected = (fin & (RK_FLAG_BAD_CHAR | RK_FLAG_TOO_LONG)) != 0;
        return (have && !is_amb && !rejected && R >= k) ? (R - k + 1) : 0u;
    };
    auto fetch_batch = [&](u32 recw, u32 pos, u32 Q, u64 (&code)[PU], RawSlot (&raw)[PU]) {
        record_codes<BITS, PU>(recw, pos, li, k, Q, code);
#pragma unroll
        for (int u = 0; u < PU; u++) raw[u] = lookup_fetch<BITS, TM>(a.db, code[u]);
    };
    auto decode_batch = [&](const u64 (&code)[PU], const RawSlot (&raw)[PU], u32 pos, u32 Q, u64 (&desc)[PU]) {
#pragma unroll
        for (int u = 0; u < PU; u++) {
            const u32 j = pos + u * G + li;
            const u64 d = lookup_decode<BITS, TM>(a.db, raw[u], code[u]);
            desc[u] = j < Q ? d : 0ull;
        }
    };

#ifdef RK_STAMPS
    unsigned long long st_[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long t_ = rk_now();
#endif
    // prologue: the first tile's inputs and its first batch of descriptors
    u32 c_recw, c_R, c_fin;
    bool c_have;
    u64 desc0[PU];
    {
        load_tile(wave_global, c_recw, c_R, c_fin, c_have);
        u64 code[PU];
        RawSlot raw[PU];
        const u32 Q0 = mer_count(c_R, c_fin, c_have);
        fetch_batch(c_recw, 0u, Q0, code, raw);
        decode_batch(code, raw, 0u, Q0, desc0);
    }

    for (u64 tile = wave_global; tile < n_tiles; tile += wave_count) {
        const bool have = c_have;
        const u64 r = have ? tile_read(a, tile * NG + gi, perm_given) : 0ull;
        const u32 fin = c_fin;
        u32 R = c_R;
        {
            const u32 cap_syms = (wpr * 32u) / BITS;
            R = R < cap_syms ? R : cap_syms;
        }
        u32 flags = fin & (RK_FLAG_BAD_CHAR | RK_FLAG_AMBIGUOUS | RK_FLAG_TOO_LONG);
        const bool is_amb = (fin & RK_FLAG_AMBIGUOUS) != 0;
        const bool rejected = (fin & (RK_FLAG_BAD_CHAR | RK_FLAG_TOO_LONG)) != 0;
        if (R < k) flags |= RK_FLAG_TOO_SHORT;
        const u32 Q = mer_count(c_R, fin, have);
        const float QT = (float)(int)Q * T;  // int * float (PlacementProcess.java:728)

        int cnt = 0;  // chunk items waiting in the list
        auto flush = [&]() {
            int wcnt = __builtin_amdgcn_readlane(cnt, 0);
            wcnt = max(wcnt, __builtin_amdgcn_readlane(cnt, 32));
            wcnt = max(wcnt, __builtin_amdgcn_readlane(cnt, 16));
            wcnt = max(wcnt, __builtin_amdgcn_readlane(cnt, 48));
            for (int i = cnt + (int)li; i < wcnt + 2 * U; i += G) items[i] = ITEM_FILLER;
            wave_lds_fence();
            if (a.db.mono) accumulate_units<G, U, true>(S, items, wcnt, li, rows_rs, QT, T);
            else accumulate_units<G, U, false>(S, items, wcnt, li, rows_rs, QT, T);
            wave_lds_fence();
            cnt = 0;
        };
        // one batch of PU*16 positions: descriptors -> unit items in k-mer order (or, for rows too long for the list, the
        // row-cursor fallback); identical to place_packed_kernel's emit phase
        auto emit_batch = [&](const u64 (&desc)[PU], bool more) {
            u32 nch[PU], excl[PU];
            int total = 0;
            bool wide_rows = false;
#pragma unroll
            for (int u = 0; u < PU; u++) {
                const u32 lenp = (u32)desc[u] & DESC_LEN_MASK;
                nch[u] = (lenp + G - 1) >> 4;
                wide_rows = wide_rows || nch[u] > 15u;
            }
            auto row_scan = [](u32 v) {  // inclusive prefix sum over the 16 lanes of a DPP row (row_shr shifts zeros in)
                v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true);
                v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true);
                v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true);
                v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true);
                return v;
            };
            if (!__any(wide_rows)) {
                // rows of <= 15 units each (240 entries: nearly always): four sub-batches share one scan, one byte each
                // (a byte's sum over the 16 lanes is <= 240, so nothing carries into its neighbour)
#pragma unroll
                for (int u0 = 0; u0 < PU; u0 += 4) {
                    u32 packed = nch[u0];
                    if (u0 + 1 < PU) packed |= nch[u0 + 1 < PU ? u0 + 1 : u0] << 8;
                    if (u0 + 2 < PU) packed |= nch[u0 + 2 < PU ? u0 + 2 : u0] << 16;
                    if (u0 + 3 < PU) packed |= nch[u0 + 3 < PU ? u0 + 3 : u0] << 24;
                    const u32 incl = row_scan(packed);
                    const u32 tot = row_bcast32<15>(incl);
#pragma unroll
                    for (int b = 0; b < 4; b++) {
                        if (u0 + b < PU) {
                            const int u = u0 + b < PU ? u0 + b : u0;
                            excl[u] = (u32)total + ((incl >> (8 * b)) & 0xFFu) - nch[u];
                            total += (int)((tot >> (8 * b)) & 0xFFu);
                        }
                    }
                }
            } else {
#pragma unroll
                for (int u = 0; u < PU; u++) {
                    const u32 incl = row_scan(nch[u]);
                    excl[u] = (u32)total + incl - nch[u];
                    total += (int)row_bcast32<15>(incl);
                }
            }
            if (__any(more && cnt + total > cap_items)) flush();
            if (__any(more && total > cap_items)) {
                const int per_part = cap_rows / G;
                for (int u_lo = 0; u_lo < PU; u_lo += per_part) {
                    int rc = 0;
#pragma unroll
                    for (int u = 0; u < PU; u++) {
                        const bool part = u >= u_lo && u < u_lo + per_part;
                        const bool hit = part && ((u32)desc[u] & DESC_LEN_MASK) != 0;
                        const u64 bal = __ballot(hit);
                        const u64 sub = group_bits<G>(bal, gi);
                        if (hit) list[rc + count_below<G>(sub, li)] = desc[u];
                        rc += __builtin_popcountll(sub);
                    }
                    if (li == 0) list[rc] = 0;
                    wave_lds_fence();
                    if (__any(rc > 0)) accumulate_list<G, U, false>(S, nb, list, rc, li, a.db.rows, QT, T);
                    wave_lds_fence();
                }
            } else {
#pragma unroll
                for (int u = 0; u < PU; u++) {
                    const u32 rb = (u32)(desc[u] >> DESC_LEN_BITS) * 8u;  // byte offset of the row (128-byte units)
                    const int base = cnt + (int)excl[u];
                    if (nch[u] > 0) items[base] = rb;
                    if (nch[u] > 1) items[base + 1] = rb + 128u;
                    for (u32 c = 2; __any(c < nch[u]); c++)
                        if (c < nch[u]) items[base + (int)c] = rb + c * 128u;
                }
                cnt += total;
            }
        };

        RK_STAMP(0);  // tile setup
        emit_batch(desc0, Q > 0);
        for (u32 pos = PU * G; __any(pos < Q); pos += PU * G) {  // reads longer than PU*16 + k - 1 symbols
            u64 code[PU], desc[PU];
            RawSlot raw[PU];
            fetch_batch(c_recw, pos, Q, code, raw);
            __builtin_amdgcn_sched_barrier(0);
            decode_batch(code, raw, pos, Q, desc);
            emit_batch(desc, pos < Q);
        }
        RK_STAMP(2);  // scans + item emission
        // next tile's inputs: in flight during this tile's accumulate phase
        load_tile(tile + wave_count, c_recw, c_R, c_fin, c_have);
        if (__any(cnt > 0)) flush();
        RK_STAMP(4);  // accumulate (incl. its fences)
        // next tile's first batch of table gathers: in flight during this tile's select phase
        u64 ncode[PU];
        RawSlot nraw[PU];
        const u32 nQ = mer_count(c_R, c_fin, c_have);
        fetch_batch(c_recw, 0u, nQ, ncode, nraw);
        __builtin_amdgcn_sched_barrier(0);
        RK_STAMP(1);  // next tile's codes + gather issue

        u64 win_key;
        const int numBest = select_topk<G>(S, nb, li, gi, (int)a.keep_at_most, list, (int)a.list_cap, win_key RK_STAMP_ARGS);
        wave_lds_fence();
        RK_STAMP(5);  // select (rest: reset)
        decode_batch(ncode, nraw, 0u, nQ, desc0);  // (before this tile's stores, so that the wait covers loads only)
        __builtin_amdgcn_sched_barrier(0);
        RK_STAMP(3);  // decode of the next tile's descriptors (waits for its gathers)
        const bool deferred = is_amb && a.has_ascii && !rejected;  // the ASCII kernel writes these
        if (have && !deferred) weigh_and_store<G>(a, r, li, numBest, win_key, flags);
        RK_STAMP(6);  // weigh + store
    }
#ifdef RK_STAMPS
    if (lane == 0 && wave_global < 4096)
        for (int i = 0; i < 16; i++) rk_stamp_buf[wave_global * 16 + i] = st_[i];
#endif
}

// ------------------------------------------------------------------------------------------------
// place_packed16w_kernel: mid-size trees (about 1 000 .. 16 000 branches, rows short enough for the compact table).
//
// A dense score vector of 4 * n_branches bytes per read leaves a CU with 4-16 reads in flight and the kernel latency-bound
// (94 Mreads/s at 3 999 branches against 340 at 999).  Here S holds one WINDOW of win_w branches at a time, so a read costs the
// LDS of a 1 000-branch tree whatever the tree's size:
//   probe + emit once: every 128-byte row unit becomes one item of the read's MAIN list, in k-mer order, tagged with the span
//     of windows its row touches (winspec, one byte per k-mer next to the compact table: first window | min(last - first, 3) << 6, 3 = to the last window);
//   for every window: the items whose span contains it are compacted (order kept) into a WORK list and applied by
//     accumulate_units with a window filter on the slot offsets (entries of other windows fall on the scratch word); then the
//     usual select over the window, whose K best are merged into the K best so far.
// Per-branch float32 order is untouched: a branch lives in exactly one window and its entries are applied in k-mer order.
// A row that straddles windows is read once per window it touches (rows are mostly runs of neighbouring branches: rare).
// Reads whose items do not fit the main list fall back to probing the read once per window (row descriptors + row cursor).
// ------------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------------
// Tiles of reads that hit the same windows (round 3).  The windowed kernels walk a tile's four reads window by window in step; the
// reads of a real batch sit each in its own clade -- one or two windows, another one for every read -- so that a window has one of the
// four at work (scripts/clade_bench.py: 41 against 102 Mreads/s on uniform reads at 19 999 branches).  Before such a launch every read
// gets a key -- the first window most of the rows of seven of its k-mers agree on -- and a counting sort by key gives the
// order the kernels take their tiles in (PlaceArgs::perm); a batch in which most reads have no three of the seven within a window
// of each other (uniform reads) keeps its order.  Results are written at the read's own index: nothing changes for the caller.
// ------------------------------------------------------------------------------------------------
constexpr int RETILE_BINS = 64;
// hist[0..63]: reads per key; hist[64] / hist[66]: sampled reads without a clade / sampled reads; hist[65]: 1 = keep the order;
// hist[67]: sampled k-mers with a row (seven a read); hist[68]: 1 = they are no more than a random read's (retile_decide_kernel)
template <int BITS>
__device__ __forceinline__ u32 retile_read_key(const PlaceArgs &a, u64 r, u32 &spread, u32 *with_row = nullptr) {
    const u32 k = a.db.k, wpr = a.words_per_read;
    u32 R = a.lens ? a.lens[r] : a.fixed_len;
    const u32 cap_syms = (wpr * 32u) / BITS;
    R = R < cap_syms ? R : cap_syms;
    const u32 fin = a.flags_in ? a.flags_in[r] : 0u;
    const bool plain = (fin & (RK_FLAG_BAD_CHAR | RK_FLAG_AMBIGUOUS | RK_FLAG_TOO_LONG)) == 0 && R >= k;
    u32 key = 0;
    spread = 0;
    if (plain) {
        const u32 Q = R - k + 1;
        const u32 *rec = a.packed + r * wpr;
        constexpr int NS = 7;
        u32 w[NS];
#pragma unroll
        for (int i = 0; i < NS; i++) {
            const u32 p_ = (u32)(((u64)(Q - 1) * (u32)i) / (NS - 1));
            w[i] = a.db.winspec[(u32)dense_index<BITS>(extract_code<BITS>(rec, wpr, p_, k), k)] & 63u;
        }
        // the window most of the seven agree on (to within one): k-mers that also occur elsewhere in the reference (their rows are
        // filed under another clade), or that every clade shares, do not move the key while a few of the seven are at home
        u32 best = 0, best_n = 0;
#pragma unroll
        for (int i = 0; i < NS; i++) {
            u32 n_ = 0;
#pragma unroll
            for (int j = 0; j < NS; j++) n_ += (w[j] != 0u && (w[i] > w[j] ? w[i] - w[j] : w[j] - w[i]) <= 1u) ? 1u : 0u;
            n_ = w[i] != 0u ? n_ : 0u;  // (0 = no row for this k-mer -- or the first range of the tree, which is not told apart: no vote)
            best = n_ > best_n ? w[i] : best;
            best_n = max(best_n, n_);
        }
        key = best;
        if (with_row) {
            u32 nz = 0;
#pragma unroll
            for (int i = 0; i < NS; i++) nz += w[i] != 0u ? 1u : 0u;  // (a row in the tree's first range counts as none: one in sixty-four)
            *with_row = nz;
        }
        spread = best_n < 3u ? 1u : 0u;  // no three of them within a window of each other: a read without a clade (uniform reads)
    }
    return key;
}
// first a look at one read in sixty-four: hist[66] = reads looked at, hist[64] = those without a clade; a batch of such reads keeps
// its order and the full pass below ends at its first instruction (a uniform batch pays a sixty-fourth of the keys)
template <int BITS>
__global__ void __launch_bounds__(256) retile_sample_kernel(PlaceArgs a, u32 *hist) {
    const u64 r = ((u64)blockIdx.x * blockDim.x + threadIdx.x) * 64u;
    u32 spread = 0, with_row = 0;
    const bool on = r < a.n_reads;
    if (on) (void)retile_read_key<BITS>(a, r, spread, &with_row);
    const u64 sp = __ballot(on && spread != 0u), al = __ballot(on);
    for (int o = 32; o > 0; o >>= 1) with_row += (u32)__shfl_down((int)with_row, o, 64);
    if ((threadIdx.x & 63u) == 0) {
        if (sp) atomicAdd(&hist[RETILE_BINS], (u32)__builtin_popcountll(sp));
        if (al) atomicAdd(&hist[RETILE_BINS + 2], (u32)__builtin_popcountll(al));
        if (with_row) atomicAdd(&hist[RETILE_BINS + 3], with_row);
    }
}
// sparse_q16: the share of sampled k-mers with a row (x 65 536) up to which the batch counts as hitting no more often than random reads do
__global__ void retile_decide_kernel(u32 *hist, u32 sparse_q16) {
    if (threadIdx.x == 0) {
        hist[RETILE_BINS + 1] = (hist[RETILE_BINS] * 2u > hist[RETILE_BINS + 2]) ? 1u : 0u;
        hist[RETILE_BINS + 4] = ((u64)hist[RETILE_BINS + 3] * 65536ull <= (u64)hist[RETILE_BINS + 2] * 7ull * (u64)sparse_q16) ? 1u : 0u;
    }
}
template <int BITS>
__global__ void __launch_bounds__(256) retile_key_kernel(PlaceArgs a, unsigned char *keys, u32 *hist) {
    if (hist[RETILE_BINS + 1]) return;  // the batch keeps its order
    __shared__ u32 h[RETILE_BINS];
    if (threadIdx.x < RETILE_BINS) h[threadIdx.x] = 0;
    __syncthreads();
    for (u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x; r < a.n_reads; r += (u64)gridDim.x * blockDim.x) {
        u32 spread;
        const u32 key = retile_read_key<BITS>(a, r, spread);
        keys[r] = (unsigned char)key;
        atomicAdd(&h[key], 1u);
    }
    __syncthreads();
    if (threadIdx.x < RETILE_BINS && h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], h[threadIdx.x]);
}
// one block: start of every key's range (cursor[0..63])
__global__ void __launch_bounds__(64) retile_scan_kernel(u32 *hist, u32 *cursor) {
    const u32 t = threadIdx.x;
    u32 v = hist[t], incl = v;
    for (int s = 1; s < 64; s <<= 1) {
        const u32 o = (u32)__shfl_up((int)incl, s, 64);
        if ((int)t >= s) incl += o;
    }
    cursor[t] = incl - v;
}
__global__ void __launch_bounds__(256) retile_scatter_kernel(u64 n_reads, const unsigned char *keys, const u32 *hist, u32 *cursor, u32 *perm) {
    __shared__ u32 cnt[RETILE_BINS], base[RETILE_BINS];
    if (hist[RETILE_BINS + 1]) return;  // the batch keeps its order (PlaceArgs::keep_order points here: perm is not read)
    for (u64 r0 = (u64)blockIdx.x * blockDim.x; r0 < n_reads; r0 += (u64)gridDim.x * blockDim.x) {  // (block-uniform trip count)
        const u64 r = r0 + threadIdx.x;
        if (threadIdx.x < RETILE_BINS) cnt[threadIdx.x] = 0;
        __syncthreads();
        u32 key = 0, rank = 0;
        if (r < n_reads) {
            key = keys[r];
            rank = atomicAdd(&cnt[key], 1u);
        }
        __syncthreads();
        if (threadIdx.x < RETILE_BINS && cnt[threadIdx.x]) base[threadIdx.x] = atomicAdd(&cursor[threadIdx.x], cnt[threadIdx.x]);
        __syncthreads();
        if (r < n_reads) perm[base[key] + rank] = (u32)r;
        __syncthreads();
    }
}

// The tiles a first kernel marked, as a list (any order: a tile's results do not depend on when it is placed); ctl[0] = their number.
__global__ void __launch_bounds__(256) compact_marks_kernel(const unsigned char *marks, u64 n_tiles, u32 *list, u32 *ctl) {
    const u32 lane = threadIdx.x & 63;
    const u64 stride = (u64)gridDim.x * blockDim.x;
    const u64 rounds = (n_tiles + stride - 1) / stride;  // (every lane takes every round: the ballots need whole waves)
    for (u64 it = 0; it < rounds; it++) {
        const u64 t = it * stride + (u64)blockIdx.x * blockDim.x + threadIdx.x;
        const bool m = t < n_tiles && marks[t] != 0;
        const u64 bal = __ballot(m);
        if (bal == 0ull) continue;
        u32 base = 0;
        if (lane == 0) base = atomicAdd(&ctl[0], (u32)__builtin_popcountll(bal));
        base = (u32)__builtin_amdgcn_readfirstlane((int)base);
        if (m) list[base + count_below<64>(bal, lane)] = (u32)t;
    }
}

template <int BITS, int U, int PU>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(BITS == 5 ? 2 : 1))) place_packed16w_kernel(PlaceArgs a) {  // (amino acids: 286 registers left to itself = one wave per SIMD; DNA: 219, and slower when asked for two)
    constexpr int G = 16, NG = 4, TM = TM_COMPACT;
    extern __shared__ u32 lds[];
    const bool perm_given = tile_order_given(a);
    const u32 lane = threadIdx.x & 63;
    const u32 wave = threadIdx.x >> 6;
    const u32 waves_per_block = blockDim.x >> 6;
    const u32 gi = lane / G, li = lane % G;
    const u32 wave_words = NG * (a.s_stride + a.main_cap + a.work_cap);
    u32 *wbase = lds + wave * wave_words;
    u32 *S = wbase + gi * a.s_stride;
    u32 *mainl = wbase + NG * a.s_stride + gi * a.main_cap;
    u32 *work = wbase + NG * (a.s_stride + a.main_cap) + gi * a.work_cap;  // also the select phase's u64 scratch (work_cap even)
    u64 *work64 = (u64 *)work;
    const u32 nb = a.db.n_branches, k = a.db.k;
    const u32 W = a.db.win_w, NWIN = a.db.n_win;
    const float T = a.db.T;
    const int K = (int)a.keep_at_most;  // <= 16 = one winner per lane (the host picks the dense kernels otherwise)
    // this lane's group inside a wave ballot (its 16 bits in place; two 32-bit halves: no 64-bit shift by a per-lane count)
    const u64 gmask_mine = ((u64)(gi >= 2u ? 0xFFFFu << ((gi - 2u) * 16u) : 0u) << 32) | (u64)(gi < 2u ? 0xFFFFu << (gi * 16u) : 0u);
    const __amdgpu_buffer_rsrc_t rows_rs = rows_resource(a.db);
    const int main_usable = (int)a.main_cap;
    const int work_usable = (int)a.work_cap - 3 * U - 2;
    const int cap_rows = (int)(a.work_cap / 2) - 1;
    const u32 wpr = a.words_per_read;  // (beyond 16 words the k-mers are read from memory: fetch_batch)

    for (u32 i = li; i < a.s_stride; i += G) S[i] = S_UNTOUCHED;
    wave_lds_fence();
#ifdef RK_STAMPS
    unsigned long long st_[16] = {0}, t_ = rk_now();
#endif

    const u64 n_tiles = (a.n_reads + NG - 1) / NG;
    const u64 wave_global = (u64)blockIdx.x * waves_per_block + wave;
    const u64 wave_count = (u64)gridDim.x * waves_per_block;

    auto load_tile = [&](u64 tile, u32 &recw, u32 &R, u32 &fin, bool &have) {
        const u64 slot = tile * NG + gi;
        have = tile < n_tiles && slot < a.n_reads;
        const u64 r = have ? tile_read(a, slot, perm_given) : 0ull;
        recw = 0; R = 0; fin = 0;
        if (have) {
            if (li < wpr) recw = a.packed[r * wpr + li];
            R = a.lens ? a.lens[r] : a.fixed_len;
            fin = a.flags_in ? a.flags_in[r] : 0u;
        }
    };
    auto mer_count = [&](u32 R, u32 fin, bool have) -> u32 {
        const u32 cap_syms = (wpr * 32u) / BITS;
        R = R < cap_syms ? R : cap_syms;
        const bool is_amb = (fin & RK_FLAG_AMBIGUOUS) != 0;
        const bool rejected = (fin & (RK_FLAG_BAD_CHAR | RK_FLAG_TOO_LONG)) != 0;
        return (have && !is_amb && !rejected && R >= k) ? (R - k + 1) : 0u;
    };
    const u32 *c_rec = a.packed;  // the current tile's record in memory (records of more than 16 words do not fit one word per lane)
    auto fetch_batch = [&](u32 recw, u32 pos, u32 Q, u64 (&code)[PU], RawSlot (&raw)[PU], u32 (&wsr)[PU]) {
        if (wpr <= 16) {
            record_codes<BITS, PU>(recw, pos, li, k, Q, code);
        } else {
#pragma unroll
            for (int u = 0; u < PU; u++) {
                const u32 j = pos + (u32)u * G + li;
                code[u] = extract_code<BITS>(c_rec, wpr, j < Q ? j : 0u, k);
            }
        }
#pragma unroll
        for (int u = 0; u < PU; u++) {
            raw[u] = lookup_fetch<BITS, TM>(a.db, code[u]);
            wsr[u] = a.db.winspec[(u32)dense_index<BITS>(code[u], k)];
        }
    };
    auto decode_batch = [&](const u64 (&code)[PU], const RawSlot (&raw)[PU], u32 pos, u32 Q, u64 (&desc)[PU]) {
#pragma unroll
        for (int u = 0; u < PU; u++) {
            const u32 j = pos + u * G + li;
            const u64 d = lookup_decode<BITS, TM>(a.db, raw[u], code[u]);
            desc[u] = j < Q ? d : 0ull;
        }
    };
    auto row_scan = [](u32 v) {  // inclusive prefix sum over the 16 lanes of a DPP row
        v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true);
        v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true);
        v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true);
        v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true);
        return v;
    };
    auto wave_max4 = [](int v) {  // largest value among the wave's four groups (v is group-uniform)
        int m = __builtin_amdgcn_readlane(v, 0);
        m = max(m, __builtin_amdgcn_readlane(v, 16));
        m = max(m, __builtin_amdgcn_readlane(v, 32));
        return max(m, __builtin_amdgcn_readlane(v, 48));
    };

    // prologue: the first tile's inputs (record word per lane, length, flags); later tiles' inputs are loaded one tile ahead.
    // (Unlike place_packed16_kernel the table gathers are NOT carried across tiles: their ~70 live registers pushed the
    // accumulate ring into scratch spills, which serialise it; with 3-16 window passes per tile the probe latency is small change.)
    u32 c_recw, c_R, c_fin;
    bool c_have;
    // Second launch behind place_packed16s_kernel / place_hash64_kernel: only the tiles they handed over.  With a list of them
    // (compact_marks_kernel) the waves take the next tile from one queue -- a handful of tiles of ~0.4 ms each on a 65 535-branch tree,
    // which a fixed tile -> wave assignment left to a few waves; every wave ends when the queue's head passes the list's length.
    const bool queued = a.only_marked != 0u && a.marked_list != nullptr;
    auto take_marked = [&]() -> u64 {
        u32 i = 0;
        if (lane == 0) i = atomicAdd(&a.marked_ctl[1], 1u);
        i = (u32)__builtin_amdgcn_readfirstlane((int)i);
        const u32 n_marked = (u32)__builtin_amdgcn_readfirstlane((int)a.marked_ctl[0]);
        return i < n_marked ? (u64)(u32)__builtin_amdgcn_readfirstlane((int)a.marked_list[i]) : n_tiles;
    };
    u64 tile = queued ? take_marked() : wave_global;
    load_tile(tile, c_recw, c_R, c_fin, c_have);

    while (tile < n_tiles) {
        if (a.only_marked && !queued) {  // (no list: every wave looks at the marks of its own tiles)
            const u32 mark = a.tile_marks[tile];
            if (!__builtin_amdgcn_readfirstlane((int)mark)) {
                tile += wave_count;
                load_tile(tile, c_recw, c_R, c_fin, c_have);
                continue;
            }
        }
        const bool have = c_have;
        const u64 r = have ? tile_read(a, tile * NG + gi, perm_given) : 0ull;
        c_rec = a.packed + r * wpr;
        const u32 fin = c_fin;
        u32 R = c_R;
        {
            const u32 cap_syms = (wpr * 32u) / BITS;
            R = R < cap_syms ? R : cap_syms;
        }
        u32 flags = fin & (RK_FLAG_BAD_CHAR | RK_FLAG_AMBIGUOUS | RK_FLAG_TOO_LONG);
        const bool is_amb = (fin & RK_FLAG_AMBIGUOUS) != 0;
        const bool rejected = (fin & (RK_FLAG_BAD_CHAR | RK_FLAG_TOO_LONG)) != 0;
        if (R < k) flags |= RK_FLAG_TOO_SHORT;
        const u32 Q = mer_count(c_R, fin, have);
        const float QT = (float)(int)Q * T;
        RK_STAMP(0);  // tile setup

        // ---- emit: every row unit of the read whose row reaches a window of [wa, wb) -> one tagged item of the main list, k-mer
        //      order.  Called with all windows first; if some read of the tile does not fit its main list, the window range is cut
        //      into 2, 4, ... ranges (as many as the largest read's unit count asks for) and each range emitted on its own (a
        //      row's items count in every range its span reaches); a range that still does not fit takes the per-window probe of
        //      window_accumulate ----
        int mcnt = 0;
        bool overflow = false;  // wave-uniform: some read of the tile does not fit its main list
        u64 touched = 0;        // windows some row of this lane's k-mers reaches (bit w), from the first call
        u32 all_items = 0;      // row units of this lane's k-mers over the whole tree (the first emit counts them even when it overflows)
        auto emit_range = [&](u32 wa, u32 wb) {
            mcnt = 0;
            overflow = false;
            all_items = 0;
            const u64 range = bits_below(wb) & ~bits_below(wa);  // windows wa .. wb - 1
            for (u32 pos = 0; __any(pos < Q); pos += PU * G) {
                u64 code[PU], desc[PU];
                RawSlot raw[PU];
                u32 ws[PU];
                fetch_batch(c_recw, pos, Q, code, raw, ws);
                __builtin_amdgcn_sched_barrier(0);
                decode_batch(code, raw, pos, Q, desc);
                u32 nch[PU], excl[PU];
                int total = 0;
#pragma unroll
                for (int u = 0; u < PU; u++) {
                    const u32 f = ws[u] & 63u, sp = ws[u] >> 6, l0 = sp == 3u ? 63u : f + sp, l = l0 < 63u ? l0 : 63u;
                    const u64 span = ((u32)desc[u] & DESC_LEN_MASK) != 0 ? (bits_below(l + 1u) & ~bits_below(f)) : 0ull;  // bits f..l
                    touched |= span;
                    nch[u] = (span & range) ? (((u32)desc[u] & DESC_LEN_MASK) + G - 1) >> 4 : 0u;  // <= 255 units (compact table)
                    all_items += nch[u];
                }
                if (overflow) continue;  // (the rest of the read is probed for its spans and its size alone)
#pragma unroll
                for (int u = 0; u < PU; u++) {
                    const u32 incl = row_scan(nch[u]);
                    excl[u] = (u32)total + incl - nch[u];
                    total += (int)row_bcast32<15>(incl);
                }
                if (__any(pos < Q && mcnt + total > main_usable)) { overflow = true; continue; }
#pragma unroll
                for (int u = 0; u < PU; u++) {
                    // item: 128-byte unit index << 8 | the row's winspec byte (first window | span << 6, span 3 = to the last window)
                    const u32 unit = (u32)(desc[u] >> DESC_LEN_BITS) >> 4, tag = ws[u];
                    const int base = mcnt + (int)excl[u];
                    if (nch[u] > 0) mainl[base] = (unit << 8) | tag;
                    if (nch[u] > 1) mainl[base + 1] = ((unit + 1u) << 8) | tag;
                    for (u32 c = 2; __any(c < nch[u]); c++)
                        if (c < nch[u]) mainl[base + (int)c] = ((unit + c) << 8) | tag;
                }
                mcnt += total;
            }
            wave_lds_fence();
        };
        u32 n_recw = 0, n_R = 0, n_fin = 0;  // next tile's inputs: loaded after the first emit, in flight during the window passes
        bool n_have = false;
        u64 tile_windows = ~0ull;
        u32 parts = 1;  // wave-uniform: a tile that does not fit its main list whole runs its windows in 2, 4, ... ranges, one emit each

        // this window's entries of the read applied to S (window w holds the branches [w * W, w * W + win_n))
        auto window_accumulate = [&](u32 w) {
            const u32 wlo = w * W;
            const u32 win_n = nb - wlo < W ? nb - wlo : W;
            const u32 wlo4p4 = wlo * 4u + 4u, w4 = win_n * 4u;
            if (!overflow) {
                // ---- this window's items, order kept, then applied ----
                int wc = 0;
                auto flushw = [&]() {
                    const int wcnt = wave_max4(wc);
                    for (int i = wc + (int)li; i < wcnt + 2 * U; i += G) work[i] = ITEM_FILLER;
                    wave_lds_fence();
                    RK_STAMP(2);  // window compaction
                    if (!(RK_ABLATE & 512)) {
                        if (a.db.mono) accumulate_units<G, U, true, true>(S, work, wcnt, li, rows_rs, QT, T, wlo4p4, w4);
                        else accumulate_units<G, U, false, true>(S, work, wcnt, li, rows_rs, QT, T, wlo4p4, w4);
                    }
                    wave_lds_fence();
                    RK_STAMP(3);  // window accumulate
#ifdef RK_STAMPS
                    st_[10] += (unsigned long long)wcnt; st_[11] += 1; st_[12] += (unsigned long long)wc;
#endif
                    wc = 0;
                };
                // (one item per lane and chunk: ranks inside a group by ballot + mbcnt; the next chunk's items are read ahead so that
                // a chunk does not wait for its own LDS read; reads past the list stay inside the wave's LDS and are masked out)
                const int mx = (RK_ABLATE & 2048) ? 0 : wave_max4(mcnt);  // (timing only: no compaction, no accumulate)
                u32 nxt = mainl[li];
                for (int base = 0; base < mx; base += G) {
                    const int i = base + (int)li;
                    const u32 it = nxt;
                    nxt = mainl[i + G];
                    const u32 sp = (it >> 6) & 3u;
                    const u32 span = sp == 3u ? 63u : sp;  // tag 3 = "to the last window"
                    const bool sel = (w - (it & 63u)) <= span && i < mcnt;  // unsigned: w below the first window wraps
                    const u64 mg = __ballot(sel) & gmask_mine;
                    const int rank = (int)__builtin_amdgcn_mbcnt_hi((u32)(mg >> 32), __builtin_amdgcn_mbcnt_lo((u32)mg, 0u));
                    if (__any(wc + G > work_usable)) flushw();
                    if (sel) work[wc + rank] = (it >> 8) << 7;  // unit index -> byte offset of the unit
                    wc += __builtin_popcount((u32)mg) + __builtin_popcount((u32)(mg >> 32));
                }
                if (__any(wc > 0)) flushw();
                RK_STAMP(2);
            } else {
                // ---- fallback: probe the read again for this window; rows that touch it go through the row cursor ----
                int rc = 0;
                auto flush_rows = [&]() {
                    if (li == 0) work64[rc] = 0ull;  // sentinel: an empty row ends the cursor
                    wave_lds_fence();
                    if (__any(rc > 0)) accumulate_list<G, U, false>(S, nb, work64, rc, li, a.db.rows, QT, T, 0u, 0xFFFFu, wlo4p4, w4);
                    wave_lds_fence();
                    rc = 0;
                };
                for (u32 pos = 0; __any(pos < Q); pos += PU * G) {
                    u64 code[PU], desc[PU];
                    RawSlot raw[PU];
                    u32 ws[PU];
                    fetch_batch(c_recw, pos, Q, code, raw, ws);
                    __builtin_amdgcn_sched_barrier(0);
                    decode_batch(code, raw, pos, Q, desc);
#pragma unroll
                    for (int u = 0; u < PU; u++) {
                        const u32 f = ws[u] & 63u, l = (ws[u] >> 6) == 3u ? 63u : f + (ws[u] >> 6);
                        const bool hit = ((u32)desc[u] & DESC_LEN_MASK) != 0 && f <= w && w <= l;
                        if (__any(rc + G > cap_rows)) flush_rows();
                        const u64 sub = group_bits<G>(__ballot(hit), gi);
                        if (hit) work64[rc + count_below<G>(sub, li)] = desc[u];
                        rc += __builtin_popcountll(sub);
                    }
                }
                flush_rows();
            }
        };
        // Phase 0, the fast pass: every window the tile reaches is accumulated, scanned into the stream heads (select_topk) and
        // reset; the K rounds run once, over the whole tree.  Phase 1, only if an entry a stream had to drop could still belong to
        // the answer (a stream would need >= 3 of the K best): the tile again, with the exact select of every window and a merge of
        // the windows' K best.  Windows no read of the tile reaches are skipped whole (their S is in its reset state): reads of one
        // clade fill one or two of up to 64 windows.  (One loop over phases and halves so that the emit, the window pass and
        // the two selects exist once in the code: the copies an unrolled structure makes cost registers.)
        u64 acc_key = 0;  // lane r < K: rank-r key (low 16 bits: 0xFFFF - tree branch id)
        bool doubt = false;
        Heads4 hd;
        heads_clear(hd);
        bool first = true;
        for (int phase = 0; phase < 2; phase++) {
            if (phase == 1) {
                if (!__any(doubt)) break;
                acc_key = 0;
            }
            for (u32 part = 0; part < parts; part++) {
                const u32 wa = part * NWIN / parts, wb = (part + 1) * NWIN / parts;
                if (parts > 1 && !(tile_windows & bits_below(wb) & ~bits_below(wa))) continue;  // nothing in this range (or an empty one)
                if (first || parts > 1) emit_range(wa, wb);  // (the exact pass of an unsplit tile finds its main list as the fast pass left it)
                if (first) {
                    first = false;
                    auto group_or = [&](u32 t) -> u32 {
                        t |= row_ror32<8>(t);
                        t |= row_ror32<4>(t);
                        t |= row_ror32<2>(t);
                        t |= row_ror32<1>(t);
                        return (u32)(__builtin_amdgcn_readlane((int)t, 0) | __builtin_amdgcn_readlane((int)t, 16) | __builtin_amdgcn_readlane((int)t, 32) | __builtin_amdgcn_readlane((int)t, 48));
                    };
                    tile_windows = ((u64)group_or((u32)(touched >> 32)) << 32) | group_or((u32)touched);
                    if (RK_ABLATE & 4096) tile_windows = ~0ull;
                    load_tile(tile + wave_count, n_recw, n_R, n_fin, n_have);
                    RK_STAMP(1);  // probe + emit
                    if (overflow) {  // the whole tree does not fit: start over in as many ranges as the largest read needs (part becomes 0 again)
                        u32 n = all_items + row_ror32<8>(all_items);
                        n += row_ror32<4>(n);
                        n += row_ror32<2>(n);
                        n += row_ror32<1>(n);
                        const u32 most = max(max((u32)__builtin_amdgcn_readlane((int)n, 0), (u32)__builtin_amdgcn_readlane((int)n, 16)),
                                             max((u32)__builtin_amdgcn_readlane((int)n, 32), (u32)__builtin_amdgcn_readlane((int)n, 48)));
                        parts = 2;
                        while (parts < NWIN && most > parts * (u32)main_usable * 7u / 8u) parts *= 2;  // (a row that spans ranges counts in each: headroom)
                        if (parts > NWIN) parts = NWIN;
                        part = ~0u;
                        continue;
                    }
                }
                for (u32 w = wa; w < wb; w++) {
                    if (!lane_bit(tile_windows, w)) continue;
                    window_accumulate(w);
                    const u32 wlo = w * W;
                    const u32 win_n = nb - wlo < W ? nb - wlo : W;
                    if (phase == 0) {
                        heads_scan_reset<G>(S, win_n, li, wlo, hd);
                        wave_lds_fence();
                        RK_STAMP(4);  // window scan + reset
                        continue;
                    }
                    // ---- exact select over the window, merged into the K best so far ----
                    u64 win_key = 0;
                    if (!(RK_ABLATE & 1024) || w + 1 == NWIN) select_topk<G>(S, win_n, li, gi, K, work64, (int)(a.work_cap / 2), win_key RK_STAMP_ARGS);  // (timing only)
                    wave_lds_fence();
                    if (win_key != 0ull) win_key -= (u64)wlo;  // window-relative branch -> tree id (low 16 bits hold 0xFFFF - branch)
                    if (K <= 8) {  // both sets fit the 16 lanes: ranks by counting over lane rotations
                        const u64 moved = ((u64)row_ror32<8>((u32)(win_key >> 32)) << 32) | row_ror32<8>((u32)win_key);  // lane r -> lane r + 8
                        const u64 comb = li < 8 ? acc_key : moved;
                        const int rank = RankAbove<G, G - 1>::run(comb, li);
                        work64[li] = 0ull;
                        wave_lds_fence();
                        if (comb != 0ull && rank < K) work64[rank] = comb;
                        wave_lds_fence();
                        acc_key = (int)li < K ? work64[li] : 0ull;
                        wave_lds_fence();
                    } else {  // up to 32 candidates through the LDS: every lane ranks its two against all, then the list is rewritten by rank
                        work64[li] = acc_key;
                        work64[G + li] = win_key;
                        wave_lds_fence();
                        int rank_a = 0, rank_w = 0;
                        for (int t = 0; t < 2 * G; t++) {
                            const u64 o = work64[t];
                            rank_a += o > acc_key ? 1 : 0;
                            rank_w += o > win_key ? 1 : 0;
                        }
                        wave_lds_fence();
                        work64[li] = 0ull;
                        wave_lds_fence();
                        if (acc_key != 0ull && rank_a < K) work64[rank_a] = acc_key;  // (keys are unique: no two candidates share a rank)
                        if (win_key != 0ull && rank_w < K) work64[rank_w] = win_key;
                        wave_lds_fence();
                        acc_key = (int)li < K ? work64[li] : 0ull;
                        wave_lds_fence();
                    }
                }
            }
            if (phase == 0) {
                heads_rounds<G>(hd, K, li, gi, acc_key, doubt);
                RK_STAMP(5);  // rounds
            }
        }
        const int numBest = __builtin_popcountll(group_bits<G>(__ballot(acc_key != 0ull), gi));
        const bool deferred = is_amb && a.has_ascii && !rejected;  // the ASCII kernel writes these
        RK_STAMP(6);  // redo of tiles in doubt
        if (have && !deferred) weigh_and_store<G>(a, r, li, numBest, acc_key, flags);
        if (queued) {
            tile = take_marked();
            load_tile(tile, c_recw, c_R, c_fin, c_have);
        } else {
            tile += wave_count;
            c_recw = n_recw; c_R = n_R; c_fin = n_fin; c_have = n_have;  // (loaded behind this tile's probe)
        }
        RK_STAMP(7);  // weigh + store
    }
#ifdef RK_STAMPS
    if (lane == 0 && wave_global < 4096)
        for (int i = 0; i < 16; i++) rk_stamp_buf[(a.only_marked ? 4096 * 16 : 0) + wave_global * 16 + i] = st_[i];
#endif
}

// ------------------------------------------------------------------------------------------------
// place_packed16s_kernel (round 3): the windowed kernel whose cost follows the READ, not the tree.
//
// place_packed16w_kernel pays per window: a compaction pass over the read's whole item list, a pipeline fill of the accumulate
// ring and a scan + reset of all win_w slots -- 64 times over on a 65 535-branch tree, for a read that touches ~1 350 entries
// whatever the tree's size.  Here, for tiles whose reads fit (one probe batch: <= 144 k-mers; no row that spans more than three
// windows; the sorted list fits):
//   * emit: the row units of a read are placed straight from the probe registers into ONE list sorted by window, k-mer order
//     kept inside a window (counts per window by LDS adds, the segments of the tile's four reads padded to a common length so
//     that window boundaries fall on the same step for the whole wave, then one returning LDS add per row and window in k-mer
//     order -- LDS operations of a wave execute in program order -- gives every row its place);
//   * one continuous accumulate stream over that list: the register ring of row loads runs across window boundaries (no pipeline
//     fill per window), the window filter is applied when an entry's step comes;
//   * every slot of S starts from the read's Q * T, a step adds (score - T) to its slot (read + add + write: the LDS float atomic is
//     serialised lane by lane on gfx950) and ORs the slot's bit into the window's TOUCHED bitmap (integer LDS atomic, full rate; slot
//     s -> bit s >> 4 of lane s & 15); at a window boundary a lane takes its mask's slots four at a time into the stream heads and
//     restores Q * T -- nothing is scanned, nothing is compacted.  Stream heads fed in any slot order keep the slot itself; equal
//     scores inside a stream are read off the heads at the end (s0 == s1, or dr == s1) and, like a dropped candidate, put the tile in
//     doubt if they could be among the K best: it then takes the stream a second time and ranks the few entries at or above the K-th
//     score exactly.
// Tiles that do not fit are marked (PlaceArgs::tile_marks) and placed by
// place_packed16w_kernel, launched behind this kernel with only_marked set.  Results are identical to the dense kernels'.
// LDS per read: S[s_stride: 16 scratch words, one per lane, then the window's slots] | list[main_cap] | work[work_cap words: the 64
// window counters and the 16 idle words of the emit; the bitmap of the stream (16 or 32 words); the 48 keys of the second pass].
// DESIGN.md section 4.1c has the measurements and what was tried.
// ------------------------------------------------------------------------------------------------
template <int BITS, int U, int PU, bool WIDE>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(BITS == 5 ? 2 : 1))) place_packed16s_kernel(PlaceArgs a) {  // (amino acids: 264 registers left to itself, one wave per SIMD)
    constexpr int G = 16, NG = 4, TM = TM_COMPACT, HALF = U / 2, TURNS = 8;
    extern __shared__ u32 lds[];
    if (!batch_is_mine(a)) return;  // (another first kernel takes this batch)
    const bool perm_given = tile_order_given(a);
    const u32 lane = threadIdx.x & 63;
    const u32 wave = threadIdx.x >> 6;
    const u32 waves_per_block = blockDim.x >> 6;
    const u32 gi = lane / G, li = lane % G;
    const u32 wave_words = NG * (a.s_stride + a.main_cap + a.work_cap);
    u32 *wbase = lds + wave * wave_words;
    u32 *S = wbase + gi * a.s_stride;
    u32 *items = wbase + NG * a.s_stride + gi * a.main_cap;
    u32 *tlw = wbase + NG * (a.s_stride + a.main_cap) + gi * a.work_cap;  // counters + idle words (emit) / touched bitmap (stream) / u64 scratch (second pass)
    u64 *work64 = (u64 *)tlw;
    const u32 nb = a.db.n_branches, k = a.db.k;
    const u32 W = a.db.win_w;
    const float T = a.db.T;
    const int K = (int)a.keep_at_most;
    const __amdgpu_buffer_rsrc_t rows_rs = rows_resource(a.db);
    const u32 wpr = a.words_per_read;  // <= 16 (the host launches place_packed16w_kernel alone for longer records)
    const int list_usable = (int)a.main_cap - 3 * U;
    u32 *bm = tlw;  // the current window's touched bitmap (words li and, windows over 512 slots, G + li are lane li's)

#ifdef RK_STAMPS
    unsigned long long st_[16] = {0}, t_ = rk_now();
#endif

    const u64 n_tiles = (a.n_reads + NG - 1) / NG;
    const u64 wave_global = (u64)blockIdx.x * waves_per_block + wave;
    const u64 wave_count = (u64)gridDim.x * waves_per_block;

    auto load_tile = [&](u64 tile, u32 &recw, u32 &R, u32 &fin, bool &have) {
        const u64 slot = tile * NG + gi;
        have = tile < n_tiles && slot < a.n_reads;
        const u64 r = have ? tile_read(a, slot, perm_given) : 0ull;
        recw = 0; R = 0; fin = 0;
        if (have) {
            if (li < wpr) recw = a.packed[r * wpr + li];
            R = a.lens ? a.lens[r] : a.fixed_len;
            fin = a.flags_in ? a.flags_in[r] : 0u;
        }
    };
    auto row_scan = [](u32 v) {  // inclusive prefix sum over the 16 lanes of a DPP row
        v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true);
        v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true);
        v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true);
        v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true);
        return v;
    };

    u32 c_recw, c_R, c_fin;
    bool c_have;
    load_tile(wave_global, c_recw, c_R, c_fin, c_have);

    for (u64 tile = wave_global; tile < n_tiles; tile += wave_count) {
        const bool have = c_have;
        const u64 r = have ? tile_read(a, tile * NG + gi, perm_given) : 0ull;
        const u32 fin = c_fin, recw = c_recw;
        u32 R = c_R;
        {
            const u32 cap_syms = (wpr * 32u) / BITS;
            R = R < cap_syms ? R : cap_syms;
        }
        u32 flags = fin & (RK_FLAG_BAD_CHAR | RK_FLAG_AMBIGUOUS | RK_FLAG_TOO_LONG);
        const bool is_amb = (fin & RK_FLAG_AMBIGUOUS) != 0;
        const bool rejected = (fin & (RK_FLAG_BAD_CHAR | RK_FLAG_TOO_LONG)) != 0;
        if (R < k) flags |= RK_FLAG_TOO_SHORT;
        const u32 Q = (have && !is_amb && !rejected && R >= k) ? (R - k + 1) : 0u;
        const float QT = (float)(int)Q * T;
        // the next tile's inputs travel while this one is worked on
        load_tile(tile + wave_count, c_recw, c_R, c_fin, c_have);
        RK_STAMP(0);  // tile setup

        // ---- probe: one batch holds the whole read ----
        bool defer = __any(Q > (u32)(PU * G));
        u32 unit[PU], pk[PU];  // pk: row units (bits 0-15) | first window (16-21) | further windows the row reaches into (22-23)
        auto NCH = [&](int u) { return pk[u] & 0xFFFFu; };
        auto WF = [&](int u) { return (pk[u] >> 16) & 63u; };
        auto WSP = [&](int u) { return pk[u] >> 22; };
        {
            u64 code[PU], desc[PU];
            RawSlot raw[PU];
            u32 ws[PU];
            record_codes<BITS, PU>(recw, 0u, li, k, Q, code);
#pragma unroll
            for (int u = 0; u < PU; u++) {
                raw[u] = lookup_fetch<BITS, TM>(a.db, code[u]);
                ws[u] = a.db.winspec[(u32)dense_index<BITS>(code[u], k)];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < PU; u++) {
                const u32 j = (u32)u * G + li;
                const u64 d = lookup_decode<BITS, TM>(a.db, raw[u], code[u]);
                desc[u] = j < Q ? d : 0ull;
                const u32 len = (u32)desc[u] & DESC_LEN_MASK;
                const u32 n_units = (len + G - 1) >> 4;
                unit[u] = (u32)(desc[u] >> DESC_LEN_BITS) >> 4;
                pk[u] = n_units ? (n_units | ((ws[u] & 0xFFu) << 16)) : 0u;
                defer = defer || n_units > 0xFFFFu || WSP(u) == 3u;  // a row over more than three windows: the other kernel's business
            }
        }
        defer = __any(defer);
        RK_STAMP(4);  // (diagnostic split of the emit: probe)
        // ---- units per window and read: tlw[w], w < 64 ----
        *(uint4 *)(tlw + 4 * li) = make_uint4(0u, 0u, 0u, 0u);
        u32 *idle = tlw + 64 + li;  // a word of the lane's own, always 0: where the lanes whose turn it is not add nothing
        *idle = 0u;
        wave_lds_fence();
        if (!defer) {
#pragma unroll
            for (int u = 0; u < PU; u++) {
                if (pk[u]) {
                    atomicAdd(&tlw[WF(u)], NCH(u));
                    if (WSP(u) >= 1u) atomicAdd(&tlw[WF(u) + 1u], NCH(u));
                    if (WSP(u) >= 2u) atomicAdd(&tlw[WF(u) + 2u], NCH(u));
                }
            }
        }
        wave_lds_fence();
        // lane t of the wave <-> window t: the common (padded) length of the window's segment, and where it starts
        u32 seg_len, seg_end;
        int total;
        {
            const u32 *t0 = wbase + NG * (a.s_stride + a.main_cap);
            const u32 c0 = t0[lane], c1 = t0[a.work_cap + lane], c2 = t0[2 * a.work_cap + lane], c3 = t0[3 * a.work_cap + lane];
            seg_len = max(max(c0, c1), max(c2, c3));
            seg_len = (seg_len + (u32)HALF - 1u) & ~((u32)HALF - 1u);  // whole half turns of the ring: a window starts at ring slot 0 or U / 2
            u32 incl = row_scan(seg_len);
            const u32 r0 = (u32)__builtin_amdgcn_readlane((int)incl, 15), r1 = (u32)__builtin_amdgcn_readlane((int)incl, 31),
                      r2 = (u32)__builtin_amdgcn_readlane((int)incl, 47);
            incl += gi >= 1 ? r0 : 0u;
            incl += gi >= 2 ? r1 : 0u;
            incl += gi >= 3 ? r2 : 0u;
            seg_end = incl;
            total = __builtin_amdgcn_readlane((int)incl, 63);
        }
        defer = defer || total > list_usable;
        if (defer) {  // wave-uniform
#ifdef RK_STAMPS
            st_[10] += 1;
#endif
            if (lane == 0) a.tile_marks[tile] = 1;
            wave_lds_fence();
            continue;
        }
        const u64 nonempty = __ballot(seg_len != 0u);
#ifdef RK_STAMPS
        st_[11] += 1; st_[12] += (unsigned long long)total;
#endif
        wave_lds_fence();  // (every lane has read the counters)
        {   // running place of every window, per read: starts at the segment's start
            u32 *t0 = wbase + NG * (a.s_stride + a.main_cap);
            const u32 st = seg_end - seg_len;
            t0[lane] = st; t0[a.work_cap + lane] = st; t0[2 * a.work_cap + lane] = st; t0[3 * a.work_cap + lane] = st;
        }
        for (int i = (int)li; i < total + 3 * U; i += G) items[i] = ITEM_FILLER;
        wave_lds_fence();
        RK_STAMP(8);  // (counts, segments, fillers)
        // ---- places: one returning add per row and window, in k-mer order (u-major, then the lanes of the group in turn: LDS
        //      operations of a wave execute in program order).  Every turn has a result register of its own, so that the sixteen
        //      adds of a slot are issued back to back (one register for all of them made every add wait for the one before);
        //      the lane then picks its own turn's result ----
#pragma unroll
        for (int u = 0; u < PU; u++) {
            const u32 nch = NCH(u), wf = WF(u), wsp = WSP(u);
            const u64 spm = __ballot(wsp != 0u);  // rows of this round that reach into the next window or two: listed there as well
            const bool spans = spm != 0ull;
            u32 pl0 = 0, pl1 = 0, pl2 = 0;
#pragma unroll 1
            for (u32 tq = 0; tq < (u32)G; tq += TURNS) {  // TURNS turns at a time: their adds are in flight together, one wait
                u32 g0[TURNS], g1[TURNS], g2[TURNS];
#pragma unroll
                for (int j = 0; j < TURNS; j++) {
                    // every lane adds -- the others 0 to a word of their own: no exec masks to set up and take down -- and the turn's
                    // lane of each read gets the row's place back
                    const bool me = li == tq + (u32)j;
                    g1[j] = 0; g2[j] = 0;
                    g0[j] = atomicAdd(me ? &tlw[wf] : idle, me ? nch : 0u);
                    // (the same turn: a window's counter sees its rows in k-mer order whichever way they came to it.  Behind a branch of
                    //  the whole wave: an LDS atomic costs its ~40 cycles with every lane masked off, too)
                    if (spm & __ballot(me)) {
                        if (me && wsp >= 1u) g1[j] = atomicAdd(&tlw[wf + 1u], nch);
                        if (me && wsp >= 2u) g2[j] = atomicAdd(&tlw[wf + 2u], nch);
                    }
                }
#pragma unroll
                for (int j = 0; j < TURNS; j++) {
                    const bool me = li == tq + (u32)j;
                    pl0 = me ? g0[j] : pl0;
                    if (spans) {
                        pl1 = me ? g1[j] : pl1;
                        pl2 = me ? g2[j] : pl2;
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            // items = byte offsets of the row's 128-byte units, at the row's place in every window it is listed in
            if (nch > 0) items[pl0] = unit[u] << 7;
            if (nch > 1) items[pl0 + 1u] = (unit[u] + 1u) << 7;
            for (u32 c = 2; __any(c < nch); c++)
                if (c < nch) items[pl0 + c] = (unit[u] + c) << 7;
            if (spans) {
                for (u32 c = 0; __any(c < nch && wsp >= 1u); c++) {
                    if (c < nch && wsp >= 1u) items[pl1 + c] = (unit[u] + c) << 7;
                    if (c < nch && wsp >= 2u) items[pl2 + c] = (unit[u] + c) << 7;
                }
            }
        }
        wave_lds_fence();
        RK_STAMP(9);  // (places + items)
        // the counters have done their work: their first words become the touched bitmap; every slot of S starts from this read's Q * T
        const u32 QTbits = __float_as_uint(QT);
        bm[li] = 0u;
        if (WIDE) bm[G + li] = 0u;
        {
            uint4 *S4 = (uint4 *)S;
            const uint4 q4 = make_uint4(QTbits, QTbits, QTbits, QTbits);
            for (u32 q = li; q < a.s_stride / 4u; q += G) S4[q] = q4;
        }
        wave_lds_fence();
        RK_STAMP(1);  // probe + emit + sort

        // ---- the stream ----
        const u32 li8 = li * 8, li4 = li * 4;
        auto issue = [&](u32 item, u32 &b, float &v) {
            const v2u32 e = __builtin_amdgcn_raw_buffer_load_b64(rows_rs, (int)(item + li8), 0, RK_ROW_AUX);
            b = e.x;
            v = __uint_as_float(e.y);
        };
        u64 acc_key = 0;
        bool doubt = false;
        Heads4 hd;
        heads_clear(hd);
        // second pass of a tile in doubt: every entry at or above the K-th score the first pass found is a CANDIDATE (the true K best
        // are among them, and they are few); a lane keeps up to three, the group's <= 48 are ranked exactly at the end
        u32 tau_o = 0;
        u64 cand0 = 0, cand1 = 0, cand2 = 0;  // (a tile is in doubt BECAUSE some lane holds three of the best: two would never do)
        bool cand_over = false;
        // the stream, once for the fast pass (PH = 0) and once more for the tiles in doubt (PH = 1): two copies of the code, each with
        // its own select at the window ends (one copy with the pass as a variable kept both passes' state alive through every loop)
        auto run_stream = [&](auto ph) {
            constexpr int PH = decltype(ph)::value;
            // ---- fast pass ----
            u32 wcur = nonempty ? (u32)__builtin_ctzll(nonempty) : 0u;
            u64 left = nonempty & (nonempty - 1);
            int bound = nonempty ? __builtin_amdgcn_readlane((int)seg_end, (int)wcur) : 0x7FFFFFFF;  // first step behind the current window
            u32 wlo = wcur * W;
            u32 wlo4p4 = wlo * 4u + 4u, w4 = (nb - wlo < W ? nb - wlo : W) * 4u;
            // Every slot of S holds the read's Q * T (written at the tile's start, restored at a window's end), so a step adds
            // (score - T) to its slot -- the reference's (Q*T + d1) + d2 ... in k-mer order -- and ORs the slot's bit into the window's
            // TOUCHED bitmap.  Slot s of the window is bit (s >> 4) of lane (s & 15)'s mask: neighbouring branches fall to different
            // lanes, and a window's end reads nothing but the touched slots.
            auto finish_window = [&]() {
                typedef typename std::conditional<WIDE, u64, u32>::type mask_t;  // (windows of <= 512 slots: 32 bits a lane)
                mask_t m = (mask_t)bm[li];
                bm[li] = 0u;
                if (WIDE) {
                    m |= (mask_t)((u64)bm[G + li] << 32);
                    bm[G + li] = 0u;
                }
                // a lane's slots go to its four streams in turn, starting with another stream in every window (short masks would
                // otherwise all land in stream 0): a rotation by wcur & 3, in two conditional steps
                const bool r1 = (wcur & 1u) != 0u, r2 = (wcur & 2u) != 0u;
                const u32 wlom15 = wlo - 15u;
                if (__any(m != 0)) do {
                    u32 sbt[4], sbv[4], val[4];
#pragma unroll
                    for (int c = 0; c < 4; c++) {  // byte offset of the slot in S (below 64 = none: the lane's scratch word)
                        const u32 bit = WIDE ? (u32)__builtin_ctzll((u64)m) : (u32)__builtin_ctz((u32)m);
                        sbt[c] = m ? (bit << 6) + li4 + 64u : li4;
                        m &= m - 1;
                    }
                    {
                        const u32 a0 = r1 ? sbt[3] : sbt[0], a1 = r1 ? sbt[0] : sbt[1], a2 = r1 ? sbt[1] : sbt[2], a3 = r1 ? sbt[2] : sbt[3];
                        sbv[0] = r2 ? a2 : a0; sbv[1] = r2 ? a3 : a1; sbv[2] = r2 ? a0 : a2; sbv[3] = r2 ? a1 : a3;
                    }
#pragma unroll
                    for (int c = 0; c < 4; c++) {
                        u32 *ps = (u32 *)((unsigned char *)S + sbv[c]);
                        val[c] = *ps;
                        *ps = QTbits;
                    }
                    if (PH == 0) {
                        const bool h0 = sbv[0] >= 64u, h1 = sbv[1] >= 64u, h2 = sbv[2] >= 64u, h3 = sbv[3] >= 64u;
                        const float f0 = h0 ? __uint_as_float(val[0]) : -INFINITY, f1 = h1 ? __uint_as_float(val[1]) : -INFINITY,
                                    f2 = h2 ? __uint_as_float(val[2]) : -INFINITY, f3 = h3 ? __uint_as_float(val[3]) : -INFINITY;
                        heads_feed<0>(hd, f0, wlom15 + (sbv[0] >> 2));  // slot id = branch + 1
                        heads_feed<1>(hd, f1, wlom15 + (sbv[1] >> 2));
                        heads_feed<2>(hd, f2, wlom15 + (sbv[2] >> 2));
                        heads_feed<3>(hd, f3, wlom15 + (sbv[3] >> 2));
                    } else {
#pragma unroll
                        for (int c = 0; c < 4; c++) {
                            const bool is = sbv[c] >= 64u && ord_f32(__uint_as_float(val[c])) >= tau_o;
                            const u64 key = make_key(val[c], wlom15 + (sbv[c] >> 2) - 1u);
                            cand_over = cand_over || (is && cand2 != 0ull);
                            cand2 = (is && cand1 != 0ull && cand2 == 0ull) ? key : cand2;
                            cand1 = (is && cand0 != 0ull && cand1 == 0ull) ? key : cand1;
                            cand0 = (is && cand0 == 0ull) ? key : cand0;
                        }
                    }
                } while (__any(m != 0));
                wave_lds_fence();
                if (left) {
                    wcur = (u32)__builtin_ctzll(left);
                    left &= left - 1;
                    bound = __builtin_amdgcn_readlane((int)seg_end, (int)wcur);
                    wlo = wcur * W;
                    wlo4p4 = wlo * 4u + 4u;
                    w4 = (nb - wlo < W ? nb - wlo : W) * 4u;
                } else {
                    bound = 0x7FFFFFFF;
                    w4 = 0u;  // nothing is inside a window any more: what is left of the ring is skipped
                }
            };
            // A step: read + add + write of the slot (gfx950's LDS float atomic is serialised lane by lane -- ~190 cycles for a wave,
            // profiles/r03_lds_atomic_rate.txt -- so it is not used), and one non-returning integer OR into the bitmap (full rate).  No
            // control flow: an entry outside the window -- a filler, or a row that reaches into the next window -- goes through the
            // lane's own scratch word and ORs nothing, so that the compiler counts the operations in flight exactly.
            auto apply_track = [&](u32 sb_raw, float sc) {
                const u32 t = sb_raw - wlo4p4;  // byte offset of the entry's slot in the window
                const bool in = t < w4;
                u32 *ps = (u32 *)((unsigned char *)S + (in ? t + 64u : li4));
                const float old = __uint_as_float(*ps);
                *ps = __float_as_uint(old + (sc - T));
                const u32 sw = t >> 2;
                const u32 word = WIDE ? ((sw & 15u) | ((sw >> 9) << 4)) : (sw & 15u);
                __hip_atomic_fetch_or(bm + (in ? word : li), in ? 1u << ((sw >> 4) & 31u) : 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            };
            u32 sb[U], it[U];  // it[u]: the item the load of ring slot u is issued with next (read from the list a half turn at a time)
            float sc[U];
            static_assert(HALF == 4, "items travel four at a time");
            {
                const uint4 i0 = *(const uint4 *)(items), i1 = *(const uint4 *)(items + 4), i2 = *(const uint4 *)(items + 8), i3 = *(const uint4 *)(items + 12);
                // (in ring order, and kept so: the waits of the steps count the loads issued after the one they need)
                const u32 first[U] = {i0.x, i0.y, i0.z, i0.w, i1.x, i1.y, i1.z, i1.w};
#pragma unroll
                for (int u = 0; u < U; u++) {
                    issue(first[u], sb[u], sc[u]);
                    __builtin_amdgcn_sched_barrier(0);
                }
                it[0] = i2.x; it[1] = i2.y; it[2] = i2.z; it[3] = i2.w;
                it[4] = i3.x; it[5] = i3.y; it[6] = i3.z; it[7] = i3.w;
            }
            // window by window; the ring runs on across the boundaries (segments are whole half turns of it: a window ends after
            // ring slot U / 2 - 1 or U - 1)
            int s0 = 0;
            auto half_turn = [&](auto first) {
                constexpr int O = decltype(first)::value ? 0 : HALF;
#pragma unroll
                for (int u = O; u < O + HALF; u++) {
                    apply_track(sb[u], sc[u]);
                    issue(it[u], sb[u], sc[u]);
                    __builtin_amdgcn_sched_barrier(0);  // (a step waits for ITS load only: without this the scheduler gathers the turn's eight waits into one)
                }
                const uint4 nx = *(const uint4 *)(items + s0 + 2 * U);  // (16-byte aligned: the list starts on a 16-byte boundary, s0 is a multiple of 4)
                it[O] = nx.x; it[O + 1] = nx.y; it[O + 2] = nx.z; it[O + 3] = nx.w;
                s0 += HALF;
            };
            // (two copies of the window's end, so that every path into a step has the ring's loads in flight in the same order:
            //  with one copy the compiler cannot tell which half a window starts in and waits for ALL loads at every turn)
            while (bound != 0x7FFFFFFF) {
                while (s0 + U <= bound) {
                    half_turn(std::true_type());
                    half_turn(std::false_type());
                }
                if (s0 < bound) {  // the window ends in the middle of a turn: the next one (never empty) starts with the other half
                    half_turn(std::true_type());
                    RK_STAMP(3);  // stream steps
                    finish_window();
                    asm volatile("; window end, ring at its half turn");  // (two different markers: the copies must not be merged again)
                    RK_STAMP(2);  // touched-slot select of the window
                    half_turn(std::false_type());  // (after the last window: four fillers -- an early way out of the loop here would be one
                                                   //  more path into its head with the ring in another order, and the waits would be for that)
                } else {
                    RK_STAMP(3);
                    finish_window();
                    asm volatile("; window end, ring at its start");
                    RK_STAMP(2);
                }
            }
        };
        // ---- fast pass ----
        run_stream(std::integral_constant<int, 0>());
        // (what is left in the ring are fillers: loads without a memory request)
        {
                u32 win_o, win_i;
                bool d0;
                u32 kth_o;
                // Equal scores inside one stream: their order is the slots', which a stream fed in any order does not keep.  Whatever
                // arrived equal to a kept score is still to be seen at the end: equal to the best -> it is the second (s0 == s1); equal
                // to the second -> it was dropped (dr == s1); and once larger scores have pushed both out they are dropped candidates,
                // which the rounds judge anyway.  The largest such score matters only if it could be among the K best.
                float tie_v = -INFINITY;
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    const bool t = hd.s1[c] != -INFINITY && (hd.s0[c] == hd.s1[c] || hd.dr[c] == hd.s1[c]);
                    tie_v = fmaxf(tie_v, t ? hd.s1[c] : -INFINITY);
                }
                const int num = heads_rounds_raw<G, true>(hd, K, li, gi, win_o, win_i, d0, &kth_o);
                acc_key = ((int)li < num) ? (((u64)win_o << 32) | (u64)(0xFFFFu - (win_i - 1u))) : 0ull;
                const bool tie = tie_v != -INFINITY && ord_f32(tie_v) >= kth_o;
                doubt = d0 || group_bits<G>(__ballot(tie), gi) != 0;
#ifdef RK_STAMPS
                st_[13] += __any(d0) ? 1 : 0;   // (diagnostic: tiles in doubt because of a dropped candidate, as against a tie)
#endif
                RK_STAMP(5);  // rounds
        }
        if (__any(doubt)) {
#ifdef RK_STAMPS
            st_[15] += 1;
#endif
            {
                const int num0 = __builtin_popcountll(group_bits<G>(__ballot(acc_key != 0ull), gi));
                u32 t_o = acc_key != 0ull ? ~(u32)(acc_key >> 32) : 0u;  // smallest ordered score among the winners = largest complement
                t_o = group_max_u32<G>(t_o);
                tau_o = num0 >= K ? ~t_o : 0u;  // (fewer than K found: everything is a candidate -- the tile then goes to the other kernel)
            }
            run_stream(std::integral_constant<int, 1>());
            {
                // the group's candidates ranked exactly (keys are unique); more than two in one lane: the other kernel's business
                work64[li] = cand0;
                work64[G + li] = cand1;
                work64[2 * G + li] = cand2;
                wave_lds_fence();
                int rank0 = 0, rank1 = 0, rank2 = 0;
                for (int t = 0; t < 3 * G; t++) {
                    const u64 o = work64[t];
                    rank0 += o > cand0 ? 1 : 0;
                    rank1 += o > cand1 ? 1 : 0;
                    rank2 += o > cand2 ? 1 : 0;
                }
                wave_lds_fence();
                work64[li] = 0ull;
                wave_lds_fence();
                if (cand0 != 0ull && rank0 < K) work64[rank0] = cand0;
                if (cand1 != 0ull && rank1 < K) work64[rank1] = cand1;
                if (cand2 != 0ull && rank2 < K) work64[rank2] = cand2;
                wave_lds_fence();
                const u64 exact = (int)li < K ? work64[li] : 0ull;
                wave_lds_fence();
                // (a group that was not in doubt keeps its first answer: its tau came from K true winners, so both agree -- but its
                //  candidates may have overflowed without that mattering)
                const bool mine_doubt = doubt;
                acc_key = mine_doubt ? exact : acc_key;
                cand_over = cand_over && mine_doubt;
            }
        }
        if (__any(cand_over)) {  // (wave-uniform) rare twice over: the tile is left to place_packed16w_kernel; S is in its reset state
            if (lane == 0) a.tile_marks[tile] = 1;
#ifdef RK_STAMPS
            st_[10] += 1;
#endif
            continue;
        }
        const int numBest = __builtin_popcountll(group_bits<G>(__ballot(acc_key != 0ull), gi));
        const bool deferred = is_amb && a.has_ascii && !rejected;  // the ASCII kernel writes these
        RK_STAMP(6);  // redo of tiles in doubt
        if (have && !deferred) weigh_and_store<G>(a, r, li, numBest, acc_key, flags);
        RK_STAMP(7);  // weigh + store
    }
#ifdef RK_STAMPS
    if (lane == 0 && wave_global < 4096)
        for (int i = 0; i < 16; i++) rk_stamp_buf[wave_global * 16 + i] = st_[i];
#endif
}

// ------------------------------------------------------------------------------------------------
// place_hash64_kernel (round 4): large trees with short rows -- a read's scores in a HASH TABLE keyed by branch.
//
// A read touches <= ~1 900 distinct branches whatever the tree's size (profiles/r04_lsize_hist.txt: C2-like rows on 7 999 / 19 999 /
// 65 535 branches, |L| p95 1 449 / 1 529 / 1 571, max 1 889 of 3 000 reads; clade-shaped reads ~500), so the windowed kernels' cost per
// WINDOW (a sorted emit, a select at every window's end, segments padded to the tile's longest) is replaced by a cost per ENTRY:
//   * one wave per read; the LDS holds keys[NS] (branch + 1, 0 = empty; a step stamp above bit 17) and vals[NS] (f32 bits, every slot
//     preset to the read's Q * T: a first touch needs no test) -- 16 KB at NS = 2 048, eight waves per CU at any tree size;
//   * probe + emit as in the dense kernels: 64 k-mers per round, one item per 128-byte row unit, in k-mer order, no window tags;
//   * a STEP applies four consecutive units, one per 16-lane row of the wave (64 entries): every lane finds or claims its branch's slot
//     with an LDS integer compare-and-swap (full rate on gfx950, unlike the LDS float add; triangular probing), then adds
//     fl(score - T) to vals[slot] by read + add + write.  A row never repeats a branch (CustomHash_v4_FastUtil81.java:78-82) but the
//     four units of a step may belong to different k-mers: the lanes stamp their slot with the step's number (one returning integer
//     LDS max on the key word) and, if any lane finds the stamp already there, the step's four units are applied one after the
//     other -- k-mer order per branch, the reference's float32 sums bit for bit (PlacementProcess.java:719-735);
//   * select: one pass over the table into the stream heads (fed in slot order = any branch order: the heads keep the branch, ties
//     inside a stream are read off the heads at the end as in place_packed16s_kernel), K rounds; a read in doubt ranks every entry at
//     or above the K-th score exactly.
// Reads that do not fit (more keys than the table takes, a probe batch of more units than the list, too many candidates) mark their
// tile (PlaceArgs::tile_marks) and place_packed16w_kernel, launched behind this kernel with only_marked set, places it: results are
// identical to the dense kernels' either way.  A tile is the four reads at slots 4t .. 4t+3 of the batch's order, taken one after
// the other by one wave.  LDS per wave: keys[NS] | vals[NS] | items[main_cap].  DESIGN.md section 4.1d.
// ------------------------------------------------------------------------------------------------
template <int BITS, int U, int NPL, int PU, int LOGS>
__global__ void __launch_bounds__(64) place_hash64_kernel(PlaceArgs a) {
    constexpr int TM = TM_COMPACT;
    constexpr u32 NS = 1u << LOGS;
    constexpr u32 KEY_BITS = 16u, KEY_MASK = (1u << KEY_BITS) - 1u;  // key = branch + 1 (the reference's ids are 16-bit chars below 65 535); the step stamp lives above it
    extern __shared__ u32 lds[];
    // LDS: [NS] the table's keys | [64] a word of every lane's own (always 0) | [NS] the table's values | [main_cap] the list
    u32 *keys = lds;
    u32 *vals = lds + NS + 64;
    u32 *items = lds + 2 * NS + 64;
    constexpr u32 VOFF = NS + 64;  // lds[VOFF + slot] = the slot's value; lds[NS + lane] = the lane's own word, for key and value operations alike
    if (__builtin_amdgcn_groupstaticsize() != 0u) __builtin_trap();  // (lds_cas_issue takes offsets from the start of the LDS)
    if (!batch_is_mine(a)) return;  // (another first kernel takes this batch)
    const bool perm_given = tile_order_given(a);
    const u32 lane = threadIdx.x & 63;
    const u32 gi = lane >> 4, li8 = (lane & 15u) * 8u;
    const u32 k = a.db.k, wpr = a.words_per_read;
    const float T = a.db.T;
    const int K = (int)a.keep_at_most;
    const __amdgpu_buffer_rsrc_t rows_rs = rows_resource(a.db);
    const int cap_items = (int)a.main_cap - 12 * U * NPL - 8;  // fillers behind the last step + the read-ahead of the ring
    const int capc = (int)(a.main_cap / 2) - 16;         // candidate keys of a read in doubt (u64), the last 16 = the winners
    const u32 key_limit = a.work_cap;                    // keys the table may hold (< NS: a probe always finds an empty slot)
#ifdef RK_STAMPS
    unsigned long long st_[16] = {0}, t_ = rk_now();
#endif
    auto wave_scan = [&](u32 v, u32 &tot) {  // inclusive prefix sum over the wave's 64 lanes + the total
        v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true);
        v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true);
        v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true);
        v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true);
        const u32 r0 = (u32)__builtin_amdgcn_readlane((int)v, 15), r1 = (u32)__builtin_amdgcn_readlane((int)v, 31);
        const u32 r2 = (u32)__builtin_amdgcn_readlane((int)v, 47), r3 = (u32)__builtin_amdgcn_readlane((int)v, 63);
        v += gi >= 1 ? r0 : 0u;
        v += gi >= 2 ? r1 : 0u;
        v += gi >= 3 ? r2 : 0u;
        tot = r0 + r1 + r2 + r3;
        return v;
    };

    const u64 n_tiles = (a.n_reads + 3) / 4;
    // As a second launch (only_marked: behind the instantiation with the small table) the waves take the tiles that one handed over from the
    // queue place_packed16w_kernel uses, clear a tile's mark and set it again if they cannot place it either
    const bool queued = a.only_marked != 0u && a.marked_list != nullptr;
    auto take_marked = [&]() -> u64 {
        u32 i = 0;
        if (lane == 0) i = atomicAdd(&a.marked_ctl[1], 1u);
        i = (u32)__builtin_amdgcn_readfirstlane((int)i);
        const u32 n_marked = (u32)__builtin_amdgcn_readfirstlane((int)a.marked_ctl[0]);
        return i < n_marked ? (u64)(u32)__builtin_amdgcn_readfirstlane((int)a.marked_list[i]) : n_tiles;
    };
    if (a.only_marked != 0u && !queued) return;  // (never launched that way)
    for (u64 tile = queued ? take_marked() : (u64)blockIdx.x; tile < n_tiles; tile = queued ? take_marked() : tile + gridDim.x) {
        if (queued && lane == 0) a.tile_marks[tile] = 0;
        bool handed = false;  // (wave-uniform) some read of the tile is left to place_packed16w_kernel
        for (u32 g = 0; g < 4u; g++) {
            const u64 slot = tile * 4 + g;
            if (slot >= a.n_reads) break;
            u64 r = tile_read(a, slot, perm_given);
            r = ((u64)(u32)__builtin_amdgcn_readfirstlane((int)(u32)(r >> 32)) << 32) | (u32)__builtin_amdgcn_readfirstlane((int)(u32)r);
            u32 R = (u32)__builtin_amdgcn_readfirstlane((int)(a.lens ? a.lens[r] : a.fixed_len));  // (wave-uniform: one read per wave)
            const u32 fin = (u32)__builtin_amdgcn_readfirstlane((int)(a.flags_in ? a.flags_in[r] : 0u));
            {
                const u32 cap_syms = (wpr * 32u) / BITS;
                R = R < cap_syms ? R : cap_syms;
            }
            u32 flags = fin & (RK_FLAG_BAD_CHAR | RK_FLAG_AMBIGUOUS | RK_FLAG_TOO_LONG);
            const bool is_amb = (fin & RK_FLAG_AMBIGUOUS) != 0;
            const bool rejected = (fin & (RK_FLAG_BAD_CHAR | RK_FLAG_TOO_LONG)) != 0;
            if (R < k) flags |= RK_FLAG_TOO_SHORT;
            const u32 Q = (!is_amb && !rejected && R >= k) ? (R - k + 1) : 0u;  // sk.getMerCount() (AmbigSequenceKnife.java:191)
            const float QT = (float)(int)Q * T;                                  // int * float (PlacementProcess.java:728)
            const u32 *rec = a.packed + r * wpr;
            // ---- the table: empty keys, every value the read's Q * T ----
            {
                uint4 *k4 = (uint4 *)keys, *v4 = (uint4 *)vals;
                const u32 qb = __float_as_uint(QT);
                const uint4 z4 = make_uint4(0u, 0u, 0u, 0u), q4 = make_uint4(qb, qb, qb, qb);
#pragma unroll
                for (u32 i = 0; i < NS / 4u / 64u; i++) {
                    k4[i * 64u + lane] = z4;
                    v4[i * 64u + lane] = q4;
                }
                keys[NS + lane] = 0u;
            }
            wave_lds_fence();
            RK_STAMP(0);  // read setup + table reset

            bool over = false;  // (wave-uniform) the read does not fit: table, list or candidates
            u32 n_keys = 0;     // (wave-uniform) an upper bound of the keys in the table: the entries applied so far (recounted when it matters)
            u32 stamp = 1;      // (wave-uniform) number of the next step
            int cnt = 0;        // items waiting in the list

            // ---- a step: this lane's entries of the step's 4 * NPL units (unit 4 p + gi of the step is entry p of the lanes of row gi).
            //      Written without divergent control flow: lanes without an entry (padding of a unit, a filler item) work on a word
            //      of their own behind the table, lanes that have found their slot repeat a compare-and-swap that changes nothing ----
            auto apply = [&](const u32 (&sbv)[NPL], const float (&scv)[NPL]) {
                if (stamp >= 0xFFFEu) over = true;  // (the stamp has 16 bits)
                if (over) return;
                // h[p]: BYTE offset of the entry's slot in the LDS (keys at 0, the lane's own word at 4 (NS + lane)).  Conditions are kept
                // as wave masks in SGPRs and combined there (hipcc turns a bool that feeds both a select and a ballot into a 0 / 1
                // register and a second compare): a round is 6 vector instructions an entry.
                u32 key[NPL], h[NPL];
                u64 act[NPL];
                const u32 own = (NS + lane) * 4u;
#pragma unroll
                for (int p = 0; p < NPL; p++) {
                    key[p] = sbv[p] >> 2;  // (branch + 1; 0 = padding of a unit, a filler item)
                    act[p] = ~mask_eq0(key[p]);
                    h[p] = mask_select(own, ((key[p] * 0x9E3779B1u) >> (32 - LOGS)) << 2, act[p]);
                }
                {   // the table takes this step only if it cannot fill up: keys so far + the step's entries (each may be a new key)
                    u32 n_act = 0;
#pragma unroll
                    for (int p = 0; p < NPL; p++) n_act += (u32)__builtin_popcountll(act[p]);
                    if (n_keys + n_act > key_limit) {
                        // the bound counts every entry as a key of its own: before a read is given up its table is counted (uniform
                        // reads: entries ~ keys, this never runs; clade-shaped ones have 1 800 entries on 500 keys and come here near their end)
                        const uint4 *kk = (const uint4 *)keys;
                        u32 mine_n = 0;
#pragma unroll 2
                        for (u32 i = 0; i < NS / 4u / 64u; i++) {
                            const uint4 kq = kk[i * 64u + lane];
                            mine_n += (kq.x ? 1u : 0u) + (kq.y ? 1u : 0u) + (kq.z ? 1u : 0u) + (kq.w ? 1u : 0u);
                        }
                        u32 tot_n;
                        (void)wave_scan(mine_n, tot_n);
                        n_keys = tot_n;
                        if (n_keys + n_act > key_limit) { over = true; return; }
                    }
                    n_keys += n_act;
                }
#ifdef RK_STAMPS
                st_[15] += 1;
#endif
                // Every lane that is still looking has failed in every round so far, so the distance of its next hop -- triangular
                // probing: 1, 2, 3 ... slots, every slot is visited -- is the round's number: one scalar for the wave.
                // Two rounds with all of the step's 64 * NPL entries; the few that are still looking after those (a table at 0.33 mean /
                // 0.65 final load: one in twenty) go on ONE per lane -- a round of the longest probe sequence then costs a quarter
                // of the instructions and a quarter of the compare-and-swaps.  (inline assembly with one wait behind a round's
                // compare-and-swaps: hipcc gave each its own s_waitcnt)
                u32 hop4 = 0;
                u64 pend[NPL], pending = 0ull;
#pragma unroll
                for (int r = 0; r < 2; r++) {
#ifdef RK_STAMPS
                    st_[14] += 1;
#endif
                    hop4 += 4u;
                    u32 old[NPL];
#pragma unroll
                    for (int p = 0; p < NPL; p++) old[p] = lds_cas_issue(h[p], 0u, key[p]);  // (the lane's own word: compares 0 with 0)
                    lds_cas_wait(old);
                    pending = 0ull;
#pragma unroll
                    for (int p = 0; p < NPL; p++) {
                        const u64 mine = mask_eq0(old[p]), ok = mine | mask_eq_lo16(old[p], key[p]);
                        pend[p] = ~ok;
                        pending |= ~ok;
                        h[p] = mask_select((h[p] + hop4) & (NS * 4u - 4u), h[p], ok);
                    }
                    if (pending == 0ull) break;
                }
                while (pending != 0ull) {
                    // every lane's first entry that is still looking (lanes without one: their own word).  All of them have failed twice:
                    // the next hop is three slots, for those of a later turn of this loop too
                    u64 taken = 0ull, sel[NPL];
                    u32 km = 0u, hm = own;
#pragma unroll
                    for (int p = 0; p < NPL; p++) {
                        sel[p] = pend[p] & ~taken;
                        taken |= pend[p];
                        km = mask_select(km, key[p], sel[p]);
                        hm = mask_select(hm, h[p], sel[p]);
                    }
                    u32 hopm = 8u;
                    u64 pm;
                    do {
#ifdef RK_STAMPS
                        st_[9] += 1;
#endif
                        hopm += 4u;
                        u32 oldm[1];
                        oldm[0] = lds_cas_issue(hm, 0u, km);
                        lds_cas_wait(oldm);
                        const u64 mine = mask_eq0(oldm[0]), ok = mine | mask_eq_lo16(oldm[0], km);
                        pm = ~ok;
                        hm = mask_select((hm + hopm) & (NS * 4u - 4u), hm, ok);
                    } while (pm != 0ull);
                    pending = 0ull;
#pragma unroll
                    for (int p = 0; p < NPL; p++) {
                        h[p] = mask_select(h[p], hm, sel[p]);
                        pend[p] &= ~sel[p];
                        pending |= pend[p];
                    }
                }
                // the stamps and the values of the step's slots in ONE round trip: the values are what the step needs when no two of its
                // units meet in a branch (the usual case), and are read again unit by unit otherwise
                // (value of slot s: VOFF words behind its key; the lane's own word serves both and stays 0)
                u64 clash = 0ull;
                u32 at[NPL];
                float v[NPL];
                {
                    u32 prev[NPL];
#pragma unroll
                    for (int p = 0; p < NPL; p++) prev[p] = atomicMax((u32 *)((unsigned char *)lds + h[p]), mask_select(0u, key[p] | (stamp << KEY_BITS), act[p]));
#pragma unroll
                    for (int p = 0; p < NPL; p++) {
                        at[p] = mask_select(own, h[p] + VOFF * 4u, act[p]);
                        v[p] = __uint_as_float(*(u32 *)((unsigned char *)lds + at[p]));
                    }
#pragma unroll
                    for (int p = 0; p < NPL; p++) {
                        clash |= mask_ge(prev[p], stamp << KEY_BITS);  // the step's own stamp (none is larger): another of its units updates the same branch
                    }
                }
                float d[NPL];
#pragma unroll
                for (int p = 0; p < NPL; p++) d[p] = scv[p] - T;
                if (clash == 0ull) {
#pragma unroll
                    for (int p = 0; p < NPL; p++) *(u32 *)((unsigned char *)lds + at[p]) = mask_select(0u, __float_as_uint(v[p] + d[p]), act[p]);
                } else {  // the step's units one after the other: k-mer order per branch
#ifdef RK_STAMPS
                    st_[12] += 1;
#endif
                    // (a round under the row's exec mask: the other rows issue nothing, padding lanes of the row add 0 to their own word)
#pragma unroll
                    for (int p = 0; p < NPL; p++) {
                        const float dz = __uint_as_float(mask_select(0u, __float_as_uint(d[p]), act[p]));
                        u32 *const ap = (u32 *)((unsigned char *)lds + at[p]);
#pragma unroll
                        for (u32 t = 0; t < 4u; t++) {
                            if (gi == t) {
                                const float vv = __uint_as_float(*ap);
                                *ap = __float_as_uint(vv + dz);
                            }
                            wave_lds_fence();
                        }
                    }
                }
                stamp += 1u;
            };
            auto issue = [&](u32 item, u32 &b, float &v) {
                const v2u32 e = __builtin_amdgcn_raw_buffer_load_b64(rows_rs, (int)(item + li8), 0, RK_ROW_AUX);
                b = e.x;
                v = __uint_as_float(e.y);
            };
            auto flush = [&]() {
                const int steps = (cnt + 4 * NPL - 1) / (4 * NPL);
                for (int i = cnt + (int)lane; i < 4 * NPL * (steps + 2 * U); i += 64) items[i] = ITEM_FILLER;
                wave_lds_fence();
                RK_STAMP(2);  // emit
                const u32 *my = items + gi;  // entry p of this lane in step s: unit my[4 (NPL s + p)]
                u32 sb[U][NPL], it[U][NPL];
                float sc[U][NPL];
#pragma unroll
                for (int u = 0; u < U; u++)
#pragma unroll
                    for (int p = 0; p < NPL; p++) issue(my[4 * (NPL * u + p)], sb[u][p], sc[u][p]);
#pragma unroll
                for (int u = 0; u < U; u++)
#pragma unroll
                    for (int p = 0; p < NPL; p++) it[u][p] = my[4 * (NPL * (U + u) + p)];
                int s0 = 0;
                while (true) {
#pragma unroll
                    for (int u = 0; u < U; u++) {
                        apply(sb[u], sc[u]);
#pragma unroll
                        for (int p = 0; p < NPL; p++) {
                            issue(it[u][p], sb[u][p], sc[u][p]);
                            it[u][p] = my[4 * (NPL * (s0 + 2 * U + u) + p)];
                        }
                    }
                    s0 += U;
                    if (s0 >= steps) break;  // what is left in the ring are fillers
                }
                wave_lds_fence();
                cnt = 0;
                RK_STAMP(3);  // accumulate
            };

            // ---- probe + emit, PU * 64 k-mers at a time ----
            for (u32 pos = 0; pos < Q && !over; pos += (u32)PU * 64u) {
                u64 code[PU];
                RawSlot raw[PU];
                u32 nch[PU], rb[PU];
#pragma unroll
                for (int u = 0; u < PU; u++) {
                    const u32 j = pos + (u32)u * 64u + lane;
                    code[u] = extract_code<BITS>(rec, wpr, j < Q ? j : 0u, k);
                }
#pragma unroll
                for (int u = 0; u < PU; u++) raw[u] = lookup_fetch<BITS, TM>(a.db, code[u]);
                __builtin_amdgcn_sched_barrier(0);  // all PU gathers are issued before any of them is decoded
#pragma unroll
                for (int u = 0; u < PU; u++) {
                    const u32 j = pos + (u32)u * 64u + lane;
                    u64 d = lookup_decode<BITS, TM>(a.db, raw[u], code[u]);
                    d = j < Q ? d : 0ull;
                    nch[u] = (((u32)d & DESC_LEN_MASK) + 15u) >> 4;
                    rb[u] = (u32)(d >> DESC_LEN_BITS) * 8u;  // byte offset of the row's first 128-byte unit
                }
                RK_STAMP(1);  // probe
                u32 incl[PU], tot[PU];
#pragma unroll
                for (int u = 0; u < PU; u++) incl[u] = wave_scan(nch[u], tot[u]);
#pragma unroll 1
                for (int u = 0; u < PU; u++) {  // (64 k-mers at a time: the list is flushed between them when it fills up; one copy of the flush's code)
                    u32 n_u = nch[0], rb_u = rb[0], in_u = incl[0], tot_u = tot[0];
#pragma unroll
                    for (int w = 1; w < PU; w++) {
                        n_u = u == w ? nch[w] : n_u; rb_u = u == w ? rb[w] : rb_u; in_u = u == w ? incl[w] : in_u; tot_u = u == w ? tot[w] : tot_u;
                    }
                    if (over) break;
                    if (cnt + (int)tot_u > cap_items && cnt > 0) flush();
                    if ((int)tot_u > cap_items) {  // rows too long for this kernel's list
                        over = true;
                        break;
                    }
                    const int base = cnt + (int)(in_u - n_u);
                    if (n_u > 0) items[base] = rb_u;
                    if (n_u > 1) items[base + 1] = rb_u + 128u;
                    for (u32 c = 2; __any(c < n_u); c++)
                        if (c < n_u) items[base + (int)c] = rb_u + c * 128u;
                    cnt += (int)tot_u;
                }
            }
            if (cnt > 0 && !over) flush();

            // ---- select: the table into the stream heads ----
            u64 acc_key = 0;
            if (!over) {
                Heads4 hd;
                heads_clear(hd);
                const uint4 *k4 = (const uint4 *)keys, *v4 = (const uint4 *)vals;
#pragma unroll 2
                for (u32 i = 0; i < NS / 4u / 64u; i++) {
                    const uint4 kq = k4[i * 64u + lane], vq = v4[i * 64u + lane];
                    heads_feed<0>(hd, kq.x ? __uint_as_float(vq.x) : -INFINITY, kq.x & KEY_MASK);  // the heads keep the slot id = branch + 1
                    heads_feed<1>(hd, kq.y ? __uint_as_float(vq.y) : -INFINITY, kq.y & KEY_MASK);
                    heads_feed<2>(hd, kq.z ? __uint_as_float(vq.z) : -INFINITY, kq.z & KEY_MASK);
                    heads_feed<3>(hd, kq.w ? __uint_as_float(vq.w) : -INFINITY, kq.w & KEY_MASK);
                }
                RK_STAMP(4);  // table scan
                // equal scores inside one stream arrive in table order, not in branch order: seen at the end (place_packed16s_kernel)
                float tie_v = -INFINITY;
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    const bool t = hd.s1[c] != -INFINITY && (hd.s0[c] == hd.s1[c] || hd.dr[c] == hd.s1[c]);
                    tie_v = fmaxf(tie_v, t ? hd.s1[c] : -INFINITY);
                }
                u32 win_o, win_i, kth_o;
                bool d0;
                const int num = heads_rounds_raw<64, true>(hd, K, lane, 0u, win_o, win_i, d0, &kth_o);
                acc_key = ((int)lane < num) ? (((u64)win_o << 32) | (u64)(0xFFFFu - (win_i - 1u))) : 0ull;
                const bool tie = tie_v != -INFINITY && ord_f32(tie_v) >= kth_o;
                RK_STAMP(5);  // rounds
                if (d0 || __any(tie)) {
                    // ---- a read in doubt: every entry at or above the K-th score is a candidate (the true K best are among them),
                    //      ranked exactly (keys are unique: they embed the branch) ----
#ifdef RK_STAMPS
                    st_[13] += 1;
#endif
                    u64 *cl = (u64 *)items, *win = cl + capc;
                    if (lane < 16u) win[lane] = 0ull;
                    int c = 0;
                    for (u32 i = 0; i < NS / 4u / 64u; i++) {
                        const uint4 kq = k4[i * 64u + lane], vq = v4[i * 64u + lane];
                        const u32 kk[4] = {kq.x, kq.y, kq.z, kq.w}, vv[4] = {vq.x, vq.y, vq.z, vq.w};
#pragma unroll
                        for (int e = 0; e < 4; e++) {
                            const bool cand = kk[e] != 0u && ord_f32(__uint_as_float(vv[e])) >= kth_o;
                            const u64 bal = __ballot(cand);
                            const int idx = c + (int)count_below<64>(bal, lane);
                            if (cand && idx < capc) cl[idx] = make_key(vv[e], (kk[e] & KEY_MASK) - 1u);
                            c += (int)__builtin_popcountll(bal);
                        }
                    }
                    wave_lds_fence();
                    if (c > capc) {
                        over = true;
                    } else {
                        for (int j = (int)lane; j < c; j += 64) {
                            const u64 mine = cl[j];
                            int rank = 0;
                            for (int t = 0; t < c; t++) rank += (cl[t] > mine) ? 1 : 0;
                            if (rank < K) win[rank] = mine;
                        }
                        wave_lds_fence();
                        acc_key = (int)lane < K ? win[lane] : 0ull;
                        wave_lds_fence();
                    }
                    RK_STAMP(6);  // exact ranking of a read in doubt
                }
            }
            if (over) {
#ifdef RK_STAMPS
                st_[10] += 1;
#endif
                handed = true;
                continue;
            }
#ifdef RK_STAMPS
            st_[11] += 1;
#endif
            const int numBest = __builtin_popcountll(__ballot(acc_key != 0ull));
            const bool deferred = is_amb && a.has_ascii && !rejected;  // the ASCII kernel writes these
            // (keep_at_most <= 16: the winners sit in the wave's first 16-lane row -- the DPP form of the weighing; the other rows idle)
            if (!deferred) weigh_and_store<16>(a, r, lane, numBest, acc_key, flags);
            RK_STAMP(7);  // weigh + store
        }
        if (handed && lane == 0) a.tile_marks[tile] = 1;
    }
#ifdef RK_STAMPS
    if (lane == 0 && blockIdx.x < 4096)
        for (int i = 0; i < 16; i++) rk_stamp_buf[(u64)blockIdx.x * 16 + i] = st_[i];
#endif
}

// ------------------------------------------------------------------------------------------------
// large trees: one WORKGROUP per read (place_wg_kernel)
//
// When S[n_branches] takes most of a CU's LDS (C5: 19 999 branches = 80 KB) a single wave per read leaves 2 waves
// per CU.  Here the NW waves of a workgroup share one score vector and every wave owns a contiguous BRANCH range:
// rows are stored sorted by branch with a 64-byte index line in front (u16 split[i-1] = entries with branch <
// floor(i*n_branches/32)), so a wave finds its slice of a row with two 2-byte loads and streams it 64 entries per
// step.  A branch belongs to exactly one wave and that wave walks the rows in k-mer order, so per-branch float32
// order is the reference's, with no barrier inside the accumulate phase.
// ------------------------------------------------------------------------------------------------
constexpr int WG_SLOTS = 4;  // rows per lane in the slice table of a wave: a probe batch holds <= 64 * WG_SLOTS rows

// this wave's slices of the rows of a probe batch (phase A of the accumulate: the hit list is read here and nowhere later)
template <bool WIDE>
struct WaveSlices {
    typedef typename OffsetT<WIDE>::type off_t;
    off_t start[WG_SLOTS];   // byte offset of the slice's first branch id
    off_t sstart[WG_SLOTS];  // byte offset of the slice's first score
    u32 nn[WG_SLOTS];
};
template <bool WIDE>
__device__ __forceinline__ void wave_slices(WaveSlices<WIDE> &ws, const u64 *list, int cnt, u32 lane, u32 q_lo, u32 q_hi, const unsigned char *rows) {
    typedef typename OffsetT<WIDE>::type off_t;
    // the slice [start, start + n) of every row of the batch (row h lives in lane h%64, slot h/64)
    // rows of an indexed image: [64-byte index line][u16 branch[len]][f32 score[len]], len a multiple of 32
    off_t (&start)[WG_SLOTS] = ws.start;
    off_t (&sstart)[WG_SLOTS] = ws.sstart;
    u32 (&nn)[WG_SLOTS] = ws.nn;
#pragma unroll
    for (int sl = 0; sl < WG_SLOTS; sl++) {
        const int h = (int)lane + 64 * sl;
        const bool have = h < cnt;
        const u64 d = list[have ? h : 0];
        const off_t off = (off_t)(d >> DESC_LEN_BITS) << 3;  // byte offset of the branch array
        const u32 lenp = (u32)d & DESC_LEN_MASK;
        const unsigned short *split = (const unsigned short *)(rows + off - 64);
        const u32 lo_raw = split[q_lo ? q_lo - 1 : 0];
        const u32 hi_raw = split[q_hi ? q_hi - 1 : 0];
        const u32 lo = q_lo ? lo_raw : 0u;
        const u32 hi = q_hi > q_lo ? hi_raw : lo;  // (a wave may own an empty range when waves outnumber the pass's splits)
        nn[sl] = have ? hi - lo : 0u;
        start[sl] = off + (off_t)lo * 2;
        sstart[sl] = off + (off_t)lenp * 2 + (off_t)lo * 4;
    }
}

template <bool WIDE, int U>
__device__ __forceinline__ void wave_accumulate(u32 *S, u32 nb, u32 base, const WaveSlices<WIDE> &ws, u32 lane, const unsigned char *rows, float QT, float T) {
    typedef typename OffsetT<WIDE>::type off_t;
    const off_t (&start)[WG_SLOTS] = ws.start;
    const off_t (&sstart)[WG_SLOTS] = ws.sstart;
    const u32 (&nn)[WG_SLOTS] = ws.nn;
    u64 mk[WG_SLOTS];
#pragma unroll
    for (int sl = 0; sl < WG_SLOTS; sl++) mk[sl] = __ballot(nn[sl] > 0);
    // ---- phase B: stream the non-empty slices in row order through the register ring (all state wave-uniform) ----
    int sl = 0;
    u64 m = mk[0];
    int rem = 0;
    off_t cur = 0, scur = 0;
    bool done = false;
    const u32 lane2 = lane * 2, lane4 = lane * 4;
    u32 br[U];
    float sc[U];
    auto gen = [&](u32 &b, float &v) {
        if (rem <= 0 && !done) {  // uniform: next non-empty slice
            while (m == 0 && sl < WG_SLOTS - 1) {
                sl++;
                m = sl == 1 ? mk[1] : (sl == 2 ? mk[2] : mk[3]);
            }
            if (m != 0) {
                const int l = __builtin_ctzll(m);
                m &= m - 1;
                const off_t st = sl == 0 ? start[0] : (sl == 1 ? start[1] : (sl == 2 ? start[2] : start[3]));
                const off_t ss = sl == 0 ? sstart[0] : (sl == 1 ? sstart[1] : (sl == 2 ? sstart[2] : sstart[3]));
                const u32 n_ = sl == 0 ? nn[0] : (sl == 1 ? nn[1] : (sl == 2 ? nn[2] : nn[3]));
                if (WIDE) {
                    const u32 lo32 = (u32)__builtin_amdgcn_readlane((int)(u32)st, l);
                    const u32 hi32 = (u32)__builtin_amdgcn_readlane((int)(u32)((u64)st >> 32), l);
                    cur = (off_t)(((u64)hi32 << 32) | lo32);
                    const u32 slo = (u32)__builtin_amdgcn_readlane((int)(u32)ss, l);
                    const u32 shi = (u32)__builtin_amdgcn_readlane((int)(u32)((u64)ss >> 32), l);
                    scur = (off_t)(((u64)shi << 32) | slo);
                } else {
                    cur = (off_t)(u32)__builtin_amdgcn_readlane((int)(u32)st, l);
                    scur = (off_t)(u32)__builtin_amdgcn_readlane((int)(u32)ss, l);
                }
                rem = __builtin_amdgcn_readlane((int)n_, l);
            } else {
                done = true;
            }
        }
        const bool ok = (int)lane < rem;
        const off_t bo = ok ? (off_t)(cur + lane2) : (off_t)0;  // lanes without an entry read the all-0xFF line 0: "skip"
        const off_t so = ok ? (off_t)(scur + lane4) : (off_t)0;
        b = *(const unsigned short *)(rows + bo);
        v = *(const float *)(rows + so);
        rem -= 64;
        cur += 128;
        scur += 256;
    };
#pragma unroll
    for (int u = 0; u < U; u++) gen(br[u], sc[u]);
    while (true) {
        const bool was_done = done;
#pragma unroll
        for (int u = 0; u < U; u++) {
            apply_entry(S, nb, br[u], sc[u], QT, T, base);
            gen(br[u], sc[u]);
        }
        if (was_done) break;  // everything issued since `done` was a skip and everything before it has been applied
    }
}

__device__ __forceinline__ u64 wave_max_u64(u64 v) {  // wave-uniform maximum of a 64-bit key: its halves in turn
    const u32 hi = wave_max_u32((u32)(v >> 32));
    const u32 lo = wave_max_u32((u32)(v >> 32) == hi ? (u32)v : 0u);
    return ((u64)hi << 32) | lo;
}
// the slots [s0, s1) of S as quads for a scan: a quad's words outside the range read as UNTOUCHED
__device__ __forceinline__ uint4 quad_in_range(const uint4 *S4, u32 q, u32 s0, u32 s1) {
    uint4 v4 = S4[q];
    const u32 i = 4 * q;
    if (i < s0 || i >= s1) v4.x = S_UNTOUCHED;
    if (i + 1 < s0 || i + 1 >= s1) v4.y = S_UNTOUCHED;
    if (i + 2 < s0 || i + 2 >= s1) v4.z = S_UNTOUCHED;
    if (i + 3 < s0 || i + 3 >= s1) v4.w = S_UNTOUCHED;
    return v4;
}
// exact top-K of the slots [s0, s1) of S by K rounds of "largest key below the previous winner" (fallback of the stream heads)
__device__ __forceinline__ int select_rounds64(const u32 *S, u32 base, u32 s0, u32 s1, u32 lane, int K, u64 &win_key) {
    const uint4 *S4 = (const uint4 *)S;
    const u32 q0 = s0 / 4, q1 = (s1 + 3) / 4;
    u64 prev = ~0ull;
    int num = 0;
    win_key = 0;
    for (int r = 0; r < K; r++) {
        u64 best = 0;
        for (u32 q = q0 + lane; q < q1; q += 64) {
            const uint4 v4 = quad_in_range(S4, q, s0, s1);
            const u32 raw[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const u32 i = 4 * q + e;
                const u64 key = raw[e] != S_UNTOUCHED ? make_key(raw[e], i + base) : 0ull;
                best = (key < prev && key > best) ? key : best;
            }
        }
        const u64 mx = wave_max_u64(best);
        if (mx == 0) break;
        if ((int)lane == r) win_key = mx;
        prev = mx;
        num++;
    }
    return num;
}

// Workgroup barrier that orders LDS traffic only: __syncthreads() also waits for the wave's global stores (s_waitcnt vmcnt(0)) --
// in place_wg_kernel that made the whole workgroup wait for the result rows of the read just placed to reach memory, which nobody
// in the workgroup reads.  Waves exchange data through the LDS alone there.
__device__ __forceinline__ void wg_barrier_lds() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int BITS, int TM, bool WIDE, int U>
__global__ void __launch_bounds__(1024) place_wg_kernel(PlaceArgs a) {
    extern __shared__ u32 lds[];
    const u32 tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, NW = blockDim.x >> 6;
    u32 *S = lds;
    u64 *list = (u64 *)(lds + a.s_stride);
    u32 *wcnt = (u32 *)(list + a.list_cap);  // [NW] hits per wave of the current probe batch
    // wave winners of every pass (level-2 select input): one pass -> the hit list, free once every wave has its slices; several -> a
    // region of their own
    u64 *cand = a.n_pass == 1 ? list : (u64 *)(wcnt + 64);
    const u32 nb = a.db.n_branches, k = a.db.k;
    const float T = a.db.T;
    const int K = (int)a.keep_at_most;
    u32 batch = blockDim.x;  // k-mer positions probed per round: every hit must fit the list / the slice table
    if (batch > a.list_cap - 1) batch = a.list_cap - 1;
    if (batch > 64 * WG_SLOTS) batch = 64 * WG_SLOTS;
    // Trees whose score vector exceeds one CU's LDS (the reference accepts ids up to 65 534, CustomHash_v4_FastUtil81.java:79,87)
    // run n_pass passes over the read, each holding the branches of 32 / n_pass index-line ranges in S; every pass repeats
    // the probe (cheap next to the rows it streams) and leaves its wave winners in `cand`; per-branch float order is untouched
    // because a branch belongs to exactly one pass and one wave.
    const u32 P = a.n_pass, span = 32u / P;
    uint4 *S4w = (uint4 *)S;
    const uint4 *S4 = (const uint4 *)S;
    const uint4 reset4 = make_uint4(S_UNTOUCHED, S_UNTOUCHED, S_UNTOUCHED, S_UNTOUCHED);
    for (u32 i = tid; i < a.s_stride; i += blockDim.x) S[i] = S_UNTOUCHED;
    wg_barrier_lds();
#ifdef RK_STAMPS
    unsigned long long st_[16] = {0}, t_ = rk_now();
#endif

    for (u64 r = blockIdx.x; r < a.n_reads; r += gridDim.x) {
        u32 R = a.lens ? a.lens[r] : a.fixed_len;
        const u32 cap_syms = (a.words_per_read * 32u) / BITS;  // never read past the packed record
        R = R < cap_syms ? R : cap_syms;
        const u32 fin = a.flags_in ? a.flags_in[r] : 0u;
        u32 flags = fin & (RK_FLAG_BAD_CHAR | RK_FLAG_AMBIGUOUS | RK_FLAG_TOO_LONG);
        const bool is_amb = (fin & RK_FLAG_AMBIGUOUS) != 0;
        const bool rejected = (fin & (RK_FLAG_BAD_CHAR | RK_FLAG_TOO_LONG)) != 0;
        if (R < k) flags |= RK_FLAG_TOO_SHORT;
        const u32 Q = (!is_amb && !rejected && R >= k) ? (R - k + 1) : 0u;
        const float QT = (float)(int)Q * T;
        const u32 *rec = a.packed + r * a.words_per_read;

        for (u32 pass = 0; pass < P; pass++) {
            const u32 p_lo = pass * span, p_hi = (pass + 1) * span;
            const u32 base = (u32)(((u64)p_lo * nb) / 32);          // first branch of the pass
            const u32 win = (u32)(((u64)p_hi * nb) / 32) - base;    // branches of the pass (<= s_stride - 1)
            const u32 q_lo = p_lo + wave * span / NW, q_hi = p_lo + (wave + 1) * span / NW;  // index-line ranges of this wave
            // the slots of S this wave owns: the branches of its index-line ranges.  Nobody else touches them, so the wave goes from its
            // accumulate straight into the select of its own segment (round 4: a barrier stood between the two and the select cut S
            // into equal parts instead -- a row of neighbouring branches keeps one or two of the waves busy, the others waited for them
            // and then scanned while those waited in turn)
            const u32 s0 = (u32)(((u64)q_lo * nb) / 32) - base, s1 = (u32)(((u64)q_hi * nb) / 32) - base;
            for (u32 pos0 = 0; pos0 < Q; pos0 += batch) {
                // ---- probe: thread <-> k-mer position; hits compacted in position order across the workgroup ----
                const u32 j = pos0 + tid;
                const bool ok = tid < batch && j < Q;
                u64 desc = 0;
                if (TM != TM_HASH) {
                    const u64 d = lookup_desc<BITS, TM>(a.db, extract_code<BITS>(rec, a.words_per_read, ok ? j : 0u, k));
                    desc = ok ? d : 0ull;
                } else if (ok) {
                    desc = lookup_desc<BITS, TM>(a.db, extract_code<BITS>(rec, a.words_per_read, j, k));
                }
                const bool hit = ((u32)desc & DESC_LEN_MASK) != 0;
                RK_STAMP(0);  // probe (record words, table)
                const u64 bal = __ballot(hit);
                if (lane == 0) wcnt[wave] = (u32)__builtin_popcountll(bal);
                wg_barrier_lds();
                u32 hbase = 0, cnt = 0;
                for (u32 w = 0; w < NW; w++) {
                    const u32 c = wcnt[w];
                    hbase += w < wave ? c : 0u;
                    cnt += c;
                }
                if (hit) list[hbase + count_below<64>(bal, lane)] = desc;
                wg_barrier_lds();
                RK_STAMP(1);  // compaction of the hits (two barriers)
                // ---- accumulate: every wave applies its branch range of every row, rows in k-mer order ----
                // every wave's slices of the batch's rows first (the only reads of the hit list), then a barrier: from here on the list
                // is free -- a wave that is through with its range leaves its winners there while others still stream
                const bool work = cnt > 0 && q_hi > q_lo;
                WaveSlices<WIDE> ws;
                if (work) wave_slices<WIDE>(ws, list, (int)cnt, lane, q_lo, q_hi, a.db.rows);
                wg_barrier_lds();
                RK_STAMP(3);  // slices of the batch's rows (index lines)
                if (work) wave_accumulate<WIDE, U>(S, win, base, ws, lane, a.db.rows, QT, T);
                RK_STAMP(2);  // accumulate (this wave's branch range)
                if (pos0 + batch < Q) wg_barrier_lds();  // (the list is rewritten by the next batch's probe)
            }

            // ---- select, level 1: every wave ranks its segment of S (stream heads as in select_topk, K rounds of wave max) ----
            Heads4 hd;
            heads_clear(hd);
            const u32 sq0 = s0 / 4, sq1 = (s1 + 3) / 4;
            for (u32 q = sq0 + lane; q < sq1; q += 64) heads_feed_quad(hd, quad_in_range(S4, q, s0, s1), q);
            u32 win_o, win_i;
            bool doubt;
            int num = heads_rounds_raw<64>(hd, K, lane, 0u, win_o, win_i, doubt);
            u64 wkey = ((int)lane < num) ? (((u64)win_o << 32) | (u64)(0xFFFFu - (win_i + base))) : 0ull;
            if (__any(doubt)) {
#ifdef RK_STAMPS
                st_[12] += 1;
#endif
                // A stream dropped an entry that could be among the K best.  The K-th key the heads did find is a lower bound of the
                // true K-th (it is a real entry), so every true winner is at or above it: one more pass over the segment keeps those
                // entries -- two a lane, in registers -- and K rounds of wave maxima rank them exactly.  A lane with more than two (or
                // a segment with fewer than K entries in the heads) takes the K passes of select_rounds64.
                u64 tau = 0ull;
                if (num == K) {
                    const u32 th = (u32)__builtin_amdgcn_readlane((int)(u32)(wkey >> 32), K - 1), tl = (u32)__builtin_amdgcn_readlane((int)(u32)wkey, K - 1);
                    tau = ((u64)th << 32) | tl;
                }
                u64 c0 = 0ull, c1 = 0ull;
                bool over = num != K;
                if (!over) {
                    for (u32 q = sq0 + lane; q < sq1; q += 64) {
                        const uint4 v4 = quad_in_range(S4, q, s0, s1);
                        const u32 raw[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
                        for (int e = 0; e < 4; e++) {
                            const u64 key = raw[e] != S_UNTOUCHED ? make_key(raw[e], 4 * q + e + base) : 0ull;
                            const bool is = key >= tau && key != 0ull;
                            over = over || (is && c1 != 0ull);
                            c1 = (is && c0 != 0ull && c1 == 0ull) ? key : c1;
                            c0 = (is && c0 == 0ull) ? key : c0;
                        }
                    }
                }
                if (__any(over)) {
#ifdef RK_STAMPS
                    st_[13] += 1;
#endif
                    num = select_rounds64(S, base, s0, s1, lane, K, wkey);
                } else {
                    num = 0;
                    wkey = 0ull;
                    for (int rr = 0; rr < K; rr++) {
                        const u64 top = wave_max_u64(c0 > c1 ? c0 : c1);
                        if (top == 0ull) break;
                        if ((int)lane == rr) wkey = top;
                        num++;
                        c0 = c0 == top ? 0ull : c0;
                        c1 = c1 == top ? 0ull : c1;
                    }
                }
            }
            if ((int)lane < K) cand[(pass * NW + wave) * K + lane] = ((int)lane < num) ? wkey : 0ull;
            // reset: whole quads inside the range, single words in the two quads it may share with its neighbours' ranges
            for (u32 q = sq0 + lane; q < sq1; q += 64) {
                if (4 * q >= s0 && 4 * q + 4 <= s1) S4w[q] = reset4;
                else
                    for (u32 e = 0; e < 4; e++)
                        if (4 * q + e >= s0 && 4 * q + e < s1) S[4 * q + e] = S_UNTOUCHED;
            }
            RK_STAMP(4);  // level-1 select of this wave's segment
            wg_barrier_lds();
            RK_STAMP(5);  // barrier behind it
        }
        // ---- select, level 2 (wave 0): exact top-K of the n_pass*NW*K wave winners, then weights and output rows ----
        if (wave == 0) {
            // K rounds of "largest key left" over the candidates, a few per lane, by wave-wide maxima of the two key halves (DPP row
            // steps + four SGPR reads; round 4 -- ranking 56 ... 112 candidates against each other by 63 lane rotations or c^2 / 64 LDS
            // reads left the other waves of the workgroup waiting for ~4 000 cycles a read); keys are unique, 0 = no candidate
            const int c = (int)(P * NW) * K;
            u64 win_key = 0ull;
            int numBest = 0;
            if (c <= 256) {  // (one pass: <= 16 waves x 16 winners)
                u64 mine[4];
#pragma unroll
                for (int q = 0; q < 4; q++) mine[q] = ((int)lane + 64 * q < c) ? cand[lane + 64 * q] : 0ull;
                for (int rr = 0; rr < K; rr++) {
                    u64 best = mine[0] > mine[1] ? mine[0] : mine[1];
                    const u64 b2 = mine[2] > mine[3] ? mine[2] : mine[3];
                    const u64 top = wave_max_u64(best > b2 ? best : b2);
                    if (top == 0ull) break;
                    if ((int)lane == rr) win_key = top;
                    numBest++;
#pragma unroll
                    for (int q = 0; q < 4; q++) mine[q] = mine[q] == top ? 0ull : mine[q];
                }
            } else {  // several branch-range passes (trees beyond one CU's LDS): the counting rank
                u64 *win = cand + ((c + 1) & ~1);
                if ((int)lane < K) win[lane] = 0ull;
                wave_lds_fence();
                rank_candidates<64>(cand, c, win, K, lane);
                win_key = ((int)lane < K) ? win[lane] : 0ull;
                numBest = __builtin_popcountll(__ballot(win_key != 0));
            }
            const bool deferred = is_amb && a.has_ascii && !rejected;  // the ASCII kernel writes these
            RK_STAMP(8);  // level-2 select (wave 0)
            // (keep_at_most <= 16: the winners sit in the wave's first 16-lane row -- the DPP form of the weighing)
            if (!deferred) weigh_and_store<16>(a, r, lane, numBest, win_key, flags);
        }
        RK_STAMP(6);  // weights + store (wave 0)
        wg_barrier_lds();
        RK_STAMP(7);  // the other waves waiting for it
#ifdef RK_STAMPS
        st_[11] += 1;
#endif
    }
#ifdef RK_STAMPS
    if (lane == 0 && blockIdx.x < 512 && (wave == 0 || wave == NW - 1))
        for (int i = 0; i < 16; i++) rk_stamp_buf[((u64)blockIdx.x * 2 + (wave ? 1 : 0)) * 16 + i] = st_[i];
#endif
}

// ------------------------------------------------------------------------------------------------
// ASCII kernel for reads that contain ambiguity characters (one wave per read, sequential k-mers).
// A1/A2/A7: AmbigSequenceKnife.java:98-272, PlacementProcess.java:1129-1236.
// ------------------------------------------------------------------------------------------------
constexpr int ASCII_LIST_CAP = 160;
constexpr int ASCII_TABLE_BYTES = 640;  // the alphabet's tables in the LDS: [320] alternatives | [16] their counts | [256] character -> state / class
constexpr int RK_ASCII_RING = 8;

struct AmbArgs {
    const unsigned char *ascii;
    const u64 *seq_off;
    const unsigned char *char_table;  // [256] state | 0x80|class | 0xFF
    const unsigned char *alt_table;   // [16][20] alternatives per class
    const unsigned char *alt_count;   // [16]
    u32 amb_mode;
    u32 max_amb;
    u32 amb_chunk;  // branches covered by the LDS Samb/Camb windows per pass (== s_stride when everything fits)
    u32 s_win;      // branches of the tree the score vector S holds per pass over the read (>= n_branches unless the tree exceeds the LDS)
};

// One ambiguous k-mer (exactly one ambiguous position p of class cls): PlacementProcess.java:1129-1236.
// Alternatives are looked up one after the other into the LDS windows Samb/Camb (branch-range passes when the tree
// does not fit one window), then folded into S.
template <int BITS, int TM>
__device__ __forceinline__ void amb_position(const PlaceArgs &a, const AmbArgs &m, u32 *S, float *Samb, u32 *Camb,
                                             u32 chunk, u64 code, u32 p, u32 cls, u32 lane, float QT, bool have_pre,
                                             const u64 (&pre)[4], u32 s_lo, u32 nb, u32 p2 = 0, u32 cls2 = 0xFFu) {
    // S, Samb and Camb hold the branches [s_lo, s_lo + nb) of the tree (the whole tree unless it exceeds the LDS: then the read
    // is walked once per window); ids outside the window are dropped right after they are loaded
    const float T = a.db.T, P = a.db.P;
    // One ambiguous position: its alternatives in table order.  Two (DNA k >= 16 only, maxAmbigPerMer = 2): word number w takes
    // alt[w % count] at EACH position, count1 * count2 words in all (AmbigSequenceKnife.java:235-260) -- the cartesian product only
    // when the counts are coprime; repeated words are looked up and counted again, as the reference does.
    const bool two = cls2 != 0xFFu;
    const u32 W1 = m.alt_count[cls], W2 = two ? m.alt_count[cls2] : 1u;
    const u32 W = W1 * W2;
    auto word = [&](u32 w) {
        u64 cw = code | place_bits(m.alt_table[cls * 20 + w % W1], BITS * p);
        if (two) cw |= place_bits(m.alt_table[cls2 * 20 + w % W2], BITS * p2);
        return cw;
    };
    auto rebase = [&](u32 &x) { const u32 xw = x - s_lo; x = (xw < nb) ? xw : 0xFFFFu; };  // (padding 0xFFFF stays 0xFFFF)
    if (!two && W <= 4 && chunk >= nb) {
        // fast path (DNA: <= 4 alternatives; one Samb/Camb window covers the tree): the alternatives' descriptors are
        // looked up together and, when every row fits the wave, their entries are loaded once into registers; the
        // per-branch updates still run alternative by alternative, in the reference's order
        u64 d[4];
#pragma unroll
        for (int w = 0; w < 4; w++) {
            if (have_pre) d[w] = pre[w];  // looked up with the block's other ambiguous positions (place_ascii_kernel)
            else d[w] = ((u32)w < W) ? lookup_desc<BITS, TM>(a.db, code | place_bits(m.alt_table[cls * 20 + w], BITS * p)) : 0ull;
        }
        const u32 maxlen = max(max((u32)d[0] & DESC_LEN_MASK, (u32)d[1] & DESC_LEN_MASK),
                               max((u32)d[2] & DESC_LEN_MASK, (u32)d[3] & DESC_LEN_MASK));
        if (maxlen <= 16) {
            // every alternative's row fits 16 lanes: lane = (alternative, entry), so the pow() of pass 1 and the log10() of
            // pass 2 are each evaluated ONCE for all alternatives together; only the cheap LDS updates run alternative by
            // alternative, in the reference's order (the float += double of :1155 rounds per alternative)
            const u32 w = lane >> 4, e = lane & 15u;
            const u64 dw = w == 0 ? d[0] : (w == 1 ? d[1] : (w == 2 ? d[2] : d[3]));
            u32 x = 0xFFFFu;
            float v = 0.0f;
            if (w < W && e < ((u32)dw & DESC_LEN_MASK)) load_entry(a.db, dw, e, x, v);
            rebase(x);
            const bool have = x != 0xFFFFu;
            const double pw = (have && m.amb_mode == RK_AMB_MEAN) ? exp10((double)v) : 0.0;
            bool own = false;  // the first alternative that lists x folds it into S (:1161-1172 / :1223-1233)
#pragma unroll
            for (u32 ww = 0; ww < 4; ww++) {  // pass 1 (:1139-1157 / :1198-1219): alternative by alternative, in the reference's order
                if (have && w == ww) {
                    const u32 c = Camb[x];
                    Camb[x] = c + 1;
                    own = c == 0;
                    if (m.amb_mode == RK_AMB_MEAN) {
                        Samb[x] = (float)((double)Samb[x] + pw);
                    } else {
                        const float cur = Samb[x];
                        Samb[x] = (c == 0 || v > cur) ? v : cur;
                    }
                }
                wave_lds_fence();
            }
            // pass 2 in ONE step (round 3; it was one fenced step per alternative): the owners are known from pass 1, and by now
            // every alternative has left its share in Samb / Camb
            u32 c_own = 0;
            float samb_own = 0.0f;
            if (own) {
                c_own = Camb[x];
                samb_own = Samb[x];
                Camb[x] = 0;
                Samb[x] = 0.0f;
            }
            if (own) {
                const u32 old = S[x + 1];  // slot layout: branch x is word x + 1
                const float base = (old == S_UNTOUCHED) ? QT : __uint_as_float(old);
                float nw;
                if (m.amb_mode == RK_AMB_MEAN) {
                    const float missing = (float)(int)(W - c_own);
                    const float pad = missing * P;
                    const float tot = samb_own + pad;
                    const float avg = tot / (float)(int)W;
                    nw = (float)((double)base + (log10((double)avg) - (double)T));
                } else {
                    const float dd = samb_own - T;
                    nw = base + dd;
                }
                S[x + 1] = __float_as_uint(nw);
            }
            wave_lds_fence();
            return;
        }
        if (maxlen <= 64) {
            u32 xb[4];
            float v[4];
#pragma unroll
            for (int w = 0; w < 4; w++) {
                xb[w] = 0xFFFFu;
                v[w] = 0.0f;
                if (lane < ((u32)d[w] & DESC_LEN_MASK)) load_entry(a.db, d[w], lane, xb[w], v[w]);
                rebase(xb[w]);
            }
            bool owner[4];
#pragma unroll
            for (int w = 0; w < 4; w++) {  // pass 1 (:1139-1157 / :1198-1219)
                owner[w] = false;
                if (xb[w] != 0xFFFFu) {
                    const u32 x = xb[w];
                    const u32 c = Camb[x];
                    Camb[x] = c + 1;
                    owner[w] = c == 0;  // the first alternative that lists x folds it into S in pass 2
                    if (m.amb_mode == RK_AMB_MEAN) {
                        Samb[x] = (float)((double)Samb[x] + exp10((double)v[w]));
                    } else {
                        const float cur = Samb[x];
                        Samb[x] = (c == 0 || v[w] > cur) ? v[w] : cur;
                    }
                }
                wave_lds_fence();
            }
#pragma unroll
            for (int w = 0; w < 4; w++) {  // pass 2 (:1161-1172 / :1223-1233), one step: a lane's owned branches are all different
                if (owner[w]) {
                    const u32 x = xb[w];
                    const u32 c = Camb[x];
                    const u32 old = S[x + 1];  // slot layout: branch x is word x + 1
                    const float base = (old == S_UNTOUCHED) ? QT : __uint_as_float(old);
                    float nw;
                    if (m.amb_mode == RK_AMB_MEAN) {
                        const float missing = (float)(int)(W - c);
                        const float pad = missing * P;
                        const float tot = Samb[x] + pad;
                        const float avg = tot / (float)(int)W;
                        nw = (float)((double)base + (log10((double)avg) - (double)T));
                    } else {
                        const float dd = Samb[x] - T;
                        nw = base + dd;
                    }
                    S[x + 1] = __float_as_uint(nw);
                    Camb[x] = 0;
                    Samb[x] = 0.0f;
                }
            }
            wave_lds_fence();
            return;
        }
    }
    for (u32 lo = 0; lo < nb; lo += chunk) {
        // pass 1: gather alternatives into Samb / Camb (sequential over alternatives)
        for (u32 w = 0; w < W; w++) {
            const u64 cw = word(w);
            u64 desc = lookup_desc<BITS, TM>(a.db, cw);
            u32 len = (u32)desc & DESC_LEN_MASK;
            if (!len) continue;
            for (u32 e = lane; e < len; e += 64) {
                u32 xb0;
                float v;
                load_entry(a.db, desc, e, xb0, v);
                rebase(xb0);
                const u32 x = xb0 - lo;
                if (xb0 == 0xFFFFu || x >= chunk) continue;  // pad / outside the window; other chunk pass (x < lo wraps around)
                u32 c = Camb[x];
                Camb[x] = c + 1;
                if (m.amb_mode == RK_AMB_MEAN) {
                    // S_amb[x] += Math.pow(10, v)  (float += double, :1155)
                    Samb[x] = (float)((double)Samb[x] + exp10((double)v));
                } else {
                    float cur = Samb[x];
                    Samb[x] = (c == 0 || v > cur) ? v : cur;  // :1212-1217
                }
            }
            wave_lds_fence();
        }
        // pass 2: fold into S; the first alternative row that lists x does it (per-branch updates are independent,
        // so the L_amb visiting order does not change any S[x])
        for (u32 w = 0; w < W; w++) {
            const u64 cw = word(w);
            u64 desc = lookup_desc<BITS, TM>(a.db, cw);
            u32 len = (u32)desc & DESC_LEN_MASK;
            if (!len) continue;
            for (u32 e = lane; e < len; e += 64) {
                u32 xb;
                float vunused;
                load_entry(a.db, desc, e, xb, vunused);
                rebase(xb);
                const u32 x = xb - lo;
                if (xb == 0xFFFFu || x >= chunk) continue;
                u32 c = Camb[x];
                if (c != 0) {
                    u32 old = S[xb + 1];
                    float base = (old == S_UNTOUCHED) ? QT : __uint_as_float(old);
                    float nw;
                    if (m.amb_mode == RK_AMB_MEAN) {
                        float missing = (float)(int)(W - c);  // :1168, all float32
                        float pad = missing * P;
                        float tot = Samb[x] + pad;
                        float avg = tot / (float)(int)W;
                        nw = (float)((double)base + (log10((double)avg) - (double)T));  // :1169
                    } else {
                        float d = Samb[x] - T;  // :1230
                        nw = base + d;
                    }
                    S[xb + 1] = __float_as_uint(nw);
                    Camb[x] = 0;
                    Samb[x] = 0.0f;
                }
            }
            wave_lds_fence();
        }
    }
}

// One wave per read.  Lane <-> k-mer position, 64 positions per block: the unambiguous positions of a block are
// looked up together and their hit rows go through the same ordered row cursor / register ring as in the packed
// kernel; an ambiguous position first flushes the pending rows (k-mer order!) and is then handled on its own.
template <int BITS, int TM, bool SOA>
__global__ void __launch_bounds__(64) place_ascii_kernel(PlaceArgs a, AmbArgs m_in) {
    extern __shared__ u32 lds[];
    const u32 lane = threadIdx.x & 63;
    const u32 nb_tree = a.db.n_branches;
    u32 *S = lds;
    // S covers the branches [s_lo, s_lo + s_win) of the tree: the whole tree in one pass over the read, or -- trees beyond the
    // LDS (the reference accepts ids up to 65 534) -- one pass per window, each leaving its K best for a final merge.
    // Samb/Camb cover `chunk` branches of that window per ambiguity pass.
    const u32 s_win = m_in.s_win < nb_tree ? m_in.s_win : nb_tree;
    const u32 n_win = (nb_tree + s_win - 1) / s_win;
    const u32 chunk = m_in.amb_chunk;
    u64 *clist = (u64 *)(lds + a.s_stride);  // ASCII_LIST_CAP slots: hit list, then candidate list of select_topk
    float *Samb = (float *)(lds + a.s_stride + 2 * ASCII_LIST_CAP);
    u32 *Camb = lds + a.s_stride + 2 * ASCII_LIST_CAP + chunk;
    const u32 k = a.db.k;
    const float T = a.db.T;
    const int cap = ASCII_LIST_CAP - 1;
    const int K = (int)a.keep_at_most;
    const bool fit32 = a.db.rows_bytes < ROWS_FIT32_LIMIT;
    const __amdgpu_buffer_rsrc_t rows_rs = rows_resource(a.db);
    for (u32 i = lane; i < a.s_stride; i += 64) S[i] = S_UNTOUCHED;
    for (u32 i = lane; i < chunk; i += 64) { Samb[i] = 0.0f; Camb[i] = 0; }
    // the alternatives of every ambiguity class ([16][20] states + [16] counts, contiguous in the alphabet block) in the LDS: an
    // ambiguous k-mer looked them up in global memory, one dependent load after the other, ten k-mers per ambiguity code (round 4)
    unsigned char *alt_l = (unsigned char *)(lds + a.s_stride + 2 * ASCII_LIST_CAP + 2 * chunk);  // [ASCII_TABLE_BYTES] behind Camb (launch_ascii_v)
    for (u32 i = lane; i < 320u; i += 64) alt_l[i] = m_in.alt_table[i];
    if (lane < 16u) alt_l[320 + lane] = m_in.alt_count[lane];
    for (u32 i = lane; i < 256u; i += 64) alt_l[336 + i] = m_in.char_table[i];  // (the character table too: a block's decode is two dependent loads otherwise)
    AmbArgs m = m_in;
    m.alt_table = alt_l;
    m.alt_count = alt_l + 320;
    m.char_table = alt_l + 336;
    wave_lds_fence();
#ifdef RK_STAMPS
    unsigned long long st_[16] = {0}, t_ = rk_now();
#endif

    for (u64 r0 = (u64)blockIdx.x * 64; r0 < a.n_reads; r0 += (u64)gridDim.x * 64) {
        // each lane inspects one read's flag; the wave then serves the flagged ones in turn
        u64 rr = r0 + lane;
        u32 f = (rr < a.n_reads) ? a.flags_in[rr] : 0u;
        bool want = (f & RK_FLAG_AMBIGUOUS) && !(f & (RK_FLAG_BAD_CHAR | RK_FLAG_TOO_LONG));
        u64 todo = __ballot(want);
        while (todo) {
            int src = __builtin_ctzll(todo);
            todo &= todo - 1;
            const u64 r = r0 + src;
            const unsigned char *s = m.ascii + m.seq_off[r];
            const u32 R = (u32)(m.seq_off[r + 1] - m.seq_off[r]);
            u32 flags = RK_FLAG_AMBIGUOUS;
            if (R < k) flags |= RK_FLAG_TOO_SHORT;
            const u32 Q = R >= k ? R - k + 1 : 0;
            const float QT = (float)(int)Q * T;
            u64 acc_key = 0;  // lane r < K: rank-r winner over the windows done so far
            for (u32 wi = 0; wi < n_win; wi++) {
                const u32 s_lo = wi * s_win;
                const u32 nb = nb_tree - s_lo < s_win ? nb_tree - s_lo : s_win;  // branches of this window
                int cnt = 0;
                bool any_long = false;  // a pending row of more than 64 entries (more than one chunk of this 64-lane group)
                auto flush = [&]() {
                    if (cnt > 0) {
                        if (!SOA && fit32 && !any_long && !(RK_ABLATE & 256)) {
                            // the usual case: every pending row is one chunk -> the buffer-addressed unit path of the packed kernel
                            // (descriptors turned into unit items in place: every lane first reads its <= 3 descriptors)
                            // (slot-offset images of trees beyond the LDS -- the windowed kernel's, up to 65 535 branches -- carry tree-wide
                            // slot offsets: the window filter of accumulate_units rebases them and sends the rest to the scratch word)
                            u64 dreg[3];
#pragma unroll
                            for (int t = 0; t < 3; t++) dreg[t] = ((int)lane + 64 * t < cnt) ? clist[lane + 64 * t] : 0ull;
                            wave_lds_fence();
                            u32 *items = (u32 *)clist;
#pragma unroll
                            for (int t = 0; t < 3; t++)
                                if ((int)lane + 64 * t < cnt)
                                    items[lane + 64 * t] = ((u32)(dreg[t] >> DESC_LEN_BITS) * 8u) | ((((u32)dreg[t] & DESC_LEN_MASK) >> 4) - 1u);
                            for (int i = cnt + (int)lane; i < cnt + 2 * RK_ASCII_RING; i += 64) items[i] = ITEM_FILLER;
                            wave_lds_fence();
                            if (n_win > 1) {
                                if (a.db.mono) accumulate_units<64, RK_ASCII_RING, true, true>(S, items, cnt, lane, rows_rs, QT, T, s_lo * 4u + 4u, nb * 4u);
                                else accumulate_units<64, RK_ASCII_RING, false, true>(S, items, cnt, lane, rows_rs, QT, T, s_lo * 4u + 4u, nb * 4u);
                            } else if (a.db.mono) accumulate_units<64, RK_ASCII_RING, true>(S, items, cnt, lane, rows_rs, QT, T);
                            else accumulate_units<64, RK_ASCII_RING, false>(S, items, cnt, lane, rows_rs, QT, T);
                            wave_lds_fence();
                            cnt = 0;
                            return;
                        }
                        if (lane == 0) clist[cnt] = 0;  // sentinel: an empty row ends the cursor
                        wave_lds_fence();
                        if (!(RK_ABLATE & 256)) {
                            if (!SOA && n_win > 1) accumulate_list<64, RK_ASCII_RING, true, SOA>(S, nb, clist, cnt, lane, a.db.rows, QT, T, s_lo, 0xFFFFu, s_lo * 4u + 4u, nb * 4u);
                            else accumulate_list<64, RK_ASCII_RING, true, SOA>(S, nb, clist, cnt, lane, a.db.rows, QT, T, s_lo, SOA ? nb : 0xFFFFu);
                        }
                        wave_lds_fence();
                        cnt = 0;
                        any_long = false;
                    }
                };
                for (u32 j0 = 0; j0 < Q; j0 += 64) {
                    const u32 j = j0 + lane;
                    const bool inr = j < Q;
                    u64 code = 0;      // ambiguous positions contribute state 0
                    u32 ambmask = 0;   // bit i <=> window position i is ambiguous
                    u32 cls_lo = 0, cls_hi = 0;  // ambiguity class at the first / the last ambiguous position of the window
                    {
                        // every lane decodes two characters of the batch's 64 + k - 1 (its own position and, for the first k - 1
                        // lanes, the one 64 further on); a k-mer's other characters come from the neighbouring lanes
                        const u32 ia = j0 + lane, ib = j0 + 64 + lane;
                        const u32 ca = ia < R ? m.char_table[s[ia]] : 0u;
                        const u32 cb = (lane + 1 < k && ib < R) ? m.char_table[s[ib]] : 0u;
                        const u32 pair = ca | (cb << 8);
                        for (u32 i = 0; i < k; i++) {
                            const u32 src = lane + i;
                            const u32 got = (u32)__shfl((int)pair, (int)(src & 63u), 64);
                            const u32 c = (src < 64u ? got : got >> 8) & 0xFFu;
                            code |= place_bits((c & 0x80) ? 0u : c, BITS * i);
                            cls_lo = ((c & 0x80) && ambmask == 0) ? (c & 0x7Fu) : cls_lo;
                            cls_hi = (c & 0x80) ? (c & 0x7Fu) : cls_hi;
                            ambmask |= ((c >> 7) & 1u) << i;
                        }
                        if (!inr) { code = 0; ambmask = 0; }
                    }
                    u64 desc = 0;
                    if (inr && ambmask == 0) desc = lookup_desc<BITS, TM>(a.db, code);
                    const bool hit = ((u32)desc & DESC_LEN_MASK) != 0;
                    // one ambiguous character puts k consecutive positions on the ambiguity path: their alternatives (<= 4 each for
                    // DNA) are looked up here for all of them at once, one lane per position, instead of one position at a time
                    u64 alt_d[4] = {0ull, 0ull, 0ull, 0ull};
                    const bool one_amb = inr && __builtin_popcount(ambmask) == 1 && m.max_amb >= 1 && m.amb_mode != RK_AMB_SKIP;
                    bool pre_ok = false;
                    if (one_amb) {
                        const u32 p = __builtin_ctz(ambmask);
                        const u32 cls = cls_lo;
                        const u32 W = m.alt_count[cls];
                        if (W <= 4) {
                            pre_ok = true;
#pragma unroll
                            for (int w = 0; w < 4; w++)
                                if ((u32)w < W) alt_d[w] = lookup_desc<BITS, TM>(a.db, code | place_bits(m.alt_table[cls * 20 + w], BITS * p));
                        }
                    }
                    u64 amb_b = __ballot(inr && ambmask != 0);
                    const u64 hit_b = __ballot(hit);
                    u32 p0 = 0;
                    RK_STAMP(1);  // decode + probes of the batch
                    while (true) {  // wave-uniform: runs of unambiguous positions separated by ambiguous ones
                        const u32 na = amb_b ? (u32)__builtin_ctzll(amb_b) : 64u;
                        const u64 hb = hit_b & bits_below((u32)na) & ~bits_below((u32)p0);
                        const int nh = __builtin_popcountll(hb);
                        if (cnt + nh > cap) flush();
                        if (lane_bit(hb, lane)) clist[cnt + count_below<64>(hb, lane)] = desc;
                        cnt += nh;
                        any_long = any_long || __any(lane_bit(hb, lane) && ((u32)desc & DESC_LEN_MASK) > 64u);
                        RK_STAMP(2);  // list building
                        if (na >= 64) break;
                        flush();  // everything before the ambiguous k-mer must be applied first
                        RK_STAMP(3);  // flush (accumulate)
                        const u32 maskA = (u32)__builtin_amdgcn_readlane((int)ambmask, (int)na);
                        const u32 clo = (u32)__builtin_amdgcn_readlane((int)(u32)code, (int)na);
                        const u32 chi = (u32)__builtin_amdgcn_readlane((int)(u32)(code >> 32), (int)na);
                        const u64 codeA = ((u64)chi << 32) | clo;
                        const u32 n_amb = (u32)__builtin_popcount(maskA);
                        if (n_amb >= 1 && n_amb <= m.max_amb && n_amb <= 2 && m.amb_mode != RK_AMB_SKIP) {
                            const u32 p = __builtin_ctz(maskA);
                            const u32 cls = (u32)__builtin_amdgcn_readlane((int)cls_lo, (int)na);
                            const u32 p2 = 31u - (u32)__builtin_clz(maskA);  // the second ambiguous position (DNA k >= 16), == p otherwise
                            const u32 cls2 = n_amb == 2 ? (u32)__builtin_amdgcn_readlane((int)cls_hi, (int)na) : 0xFFu;
                            const bool have_pre = n_amb == 1 && __builtin_amdgcn_readlane((int)pre_ok, (int)na) != 0;
                            u64 pre[4];
#pragma unroll
                            for (int w = 0; w < 4; w++)
                                pre[w] = ((u64)(u32)__builtin_amdgcn_readlane((int)(u32)(alt_d[w] >> 32), (int)na) << 32) |
                                         (u32)__builtin_amdgcn_readlane((int)(u32)alt_d[w], (int)na);
                            if (!(RK_ABLATE & 128)) amb_position<BITS, TM>(a, m, S, Samb, Camb, chunk, codeA, p, cls, lane, QT, have_pre, pre, s_lo, nb, p2, cls2);
                        }
                        RK_STAMP(4);  // ambiguous position
                        amb_b &= amb_b - 1;
                        p0 = na + 1;
                        if (p0 >= 64) break;
                    }
                }
                flush();
                wave_lds_fence();
                u64 win_key;
                RK_STAMP(3);  // last flush
                select_topk<64>(S, nb, lane, 0u, K, clist, ASCII_LIST_CAP, win_key RK_STAMP_ARGS);
                RK_STAMP(5);
                wave_lds_fence();
                if (win_key != 0) win_key -= s_lo;  // the key's low 16 bits hold 0xFFFF - branch: window-relative -> tree id
                if (n_win == 1) {
                    acc_key = win_key;
                } else {  // merge this window's K best with the K best so far (keys are unique: they embed the branch id)
                    if (lane < 16) { clist[lane] = acc_key; clist[16 + lane] = win_key; }
                    if (lane < 16) clist[64 + lane] = 0ull;
                    wave_lds_fence();
                    rank_candidates<64>(clist, 32, clist + 64, K, lane);
                    acc_key = ((int)lane < K) ? clist[64 + lane] : 0ull;
                    wave_lds_fence();
                }
            }
            const int numBest = __builtin_popcountll(__ballot(acc_key != 0));
            weigh_and_store<64>(a, r, lane, numBest, acc_key, flags);
            RK_STAMP(6);
#ifdef RK_STAMPS
            st_[10] += 1;
#endif
        }
        RK_STAMP(0);  // flag inspection, idle lanes
    }
#ifdef RK_STAMPS
    if (lane == 0 && blockIdx.x < 4096)
        for (int i = 0; i < 16; i++) rk_stamp_buf[blockIdx.x * 16 + i] = st_[i];
#endif
}

// ------------------------------------------------------------------------------------------------
// count_work_kernel (round 4; diagnostic, never part of a placement): what a batch of packed reads asks of the database -- k-mers
// probed (sk.getMerCount() per read, AmbigSequenceKnife.java:191), k-mers with a row (hash.getPairsOfTopPosition2(word) != null,
// PlacementProcess.java:705-707) and the (branch, score) pairs of those rows (the loop of :719-735).  Reads the placement kernels skip
// (BAD_CHAR / TOO_LONG, shorter than k) or leave to place_ascii_kernel (AMBIGUOUS) count nothing.  One wave per read, a lane per
// k-mer; a row's true length = its padded length minus the padding that trails it (any image layout: load_entry).
// ------------------------------------------------------------------------------------------------
template <int BITS, int TM>
__global__ void __launch_bounds__(256) count_work_kernel(DbView db, const u32 *packed, u32 wpr, const u32 *lens, u32 fixed_len, const u32 *flags_in,
                                                         u64 n_reads, unsigned long long *out3) {
    const u32 lane = threadIdx.x & 63;
    const u64 wave_global = ((u64)blockIdx.x * blockDim.x + threadIdx.x) >> 6, wave_count = ((u64)gridDim.x * blockDim.x) >> 6;
    const u32 k = db.k;
    unsigned long long probed = 0, hits = 0, entries = 0;  // (per lane; summed over the wave at the end)
    for (u64 r = wave_global; r < n_reads; r += wave_count) {
        u32 R = lens ? lens[r] : fixed_len;
        const u32 cap_syms = (wpr * 32u) / BITS;
        R = R < cap_syms ? R : cap_syms;
        const u32 fin = flags_in ? flags_in[r] : 0u;
        const bool skip = (fin & (RK_FLAG_BAD_CHAR | RK_FLAG_TOO_LONG | RK_FLAG_AMBIGUOUS)) != 0 || R < k;
        const u32 Q = skip ? 0u : R - k + 1;
        const u32 *rec = packed + r * wpr;
        for (u32 j0 = 0; j0 < Q; j0 += 64) {
            const u32 j = j0 + lane;
            if (j >= Q) continue;
            probed++;
            const u64 desc = lookup_desc<BITS, TM>(db, extract_code<BITS>(rec, wpr, j, k));
            u32 len = (u32)desc & DESC_LEN_MASK;
            if (!len) continue;
            hits++;
            while (len) {  // padding only ever trails a row
                u32 br;
                float sc;
                load_entry(db, desc, len - 1, br, sc);
                if (br != 0xFFFFu) break;
                len--;
            }
            entries += len;
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        probed += __shfl_down(probed, o, 64);
        hits += __shfl_down(hits, o, 64);
        entries += __shfl_down(entries, o, 64);
    }
    if (lane == 0) {
        if (probed) atomicAdd(&out3[0], probed);
        if (hits) atomicAdd(&out3[1], hits);
        if (entries) atomicAdd(&out3[2], entries);
    }
}

// ------------------------------------------------------------------------------------------------
// pack kernel: ASCII -> packed records + lens + flags (AmbigSequenceKnife.java:103-130 char -> state)
// one thread per output word
// ------------------------------------------------------------------------------------------------
template <int BITS>
__global__ void __launch_bounds__(256) pack_reads_kernel(const unsigned char *ascii, const u64 *seq_off, u64 n_reads,
                                                          u32 words_per_read, const unsigned char *char_table, u32 k,
                                                          u32 *packed, u32 *lens, u32 *flags) {
    __shared__ unsigned char tab[256];
    tab[threadIdx.x & 255] = char_table[threadIdx.x & 255];
    __syncthreads();
    const u64 total = n_reads * words_per_read;
    for (u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (u64)gridDim.x * blockDim.x) {
        const u64 r = t / words_per_read;
        const u32 w = (u32)(t - r * words_per_read);
        const u64 o0 = seq_off[r];
        const u64 Rfull = seq_off[r + 1] - o0;
        const u32 cap_syms = (words_per_read * 32u) / BITS;
        const u32 R = Rfull > cap_syms ? cap_syms : (u32)Rfull;
        u32 fl = 0;
        if (w == 0) {
            lens[r] = R;
            if (Rfull > cap_syms) fl |= RK_FLAG_TOO_LONG;
            if (R < k) fl |= RK_FLAG_TOO_SHORT;
        }
        // symbols overlapping bits [32w, 32w+32)
        const u32 s_lo = (32u * w) / BITS;
        u32 s_hi = (32u * w + 31u) / BITS;  // inclusive
        // 32-bit arithmetic on purpose.  The first version accumulated in 64 bits (`acc |= (u64)st >> (-shift)` for the symbol
        // that straddles the word's lower edge); on gfx950 that form -- hipcc emits v_lshlrev_b64 / v_lshrrev_b64 back to back on
        // the same register pair -- lost the straddling symbol's bits in a few percent of the waves of every workgroup after the
        // first 256, differently from run to run (scripts/ubench/pack_repro.hip reproduces it in plain HIP; 5-bit records only:
        // 2-bit symbols never straddle).  Small batches never showed it, which is why it survived round 1.
        u32 acc = 0;
        const u32 wbase = 32u * w;
        for (u32 sidx = s_lo; sidx <= s_hi; sidx++) {
            if (sidx >= R) break;
            u32 c = tab[ascii[o0 + sidx]];
            u32 st = c;
            if (c == 0xFF) { fl |= RK_FLAG_BAD_CHAR; st = 0; }
            else if (c & 0x80) { fl |= RK_FLAG_AMBIGUOUS; st = 0; }
            const u32 lo = sidx * BITS;
            acc |= lo >= wbase ? st << (lo - wbase) : st >> (wbase - lo);
        }
        packed[t] = acc;
        if (fl) atomicOr(&flags[r], fl);
    }
}

}  // namespace rk
