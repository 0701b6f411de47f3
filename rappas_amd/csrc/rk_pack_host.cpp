// rk_pack_host.cpp -- the host-side read packer behind rk_pack_reads_host and rk_place_batch (host-only translation unit).
//
// What it restates: the char -> state part of AmbigSequenceKnife.initTables (AmbigSequenceKnife.java:103-130) with the alphabets of
// DNAStatesShifted.java:45-96,182-243 / AAStates.java:23-34,97-123 as rk_engine.hip:build_alphabet tabulates them.  Output is word
// for word what pack_reads_kernel (rk_kernels.hip) writes: symbol i at bits [i*B, (i+1)*B) of a little-endian bit string, B = 2
// (DNA) or 5 (amino acids); ambiguous and unsupported characters pack as state 0 and raise RK_FLAG_AMBIGUOUS / RK_FLAG_BAD_CHAR.
//
// 32 symbols are exactly 2 (DNA) or 5 (amino-acid) 32-bit words, so a read is packed in independent blocks of 32 characters:
//   * vector path (AVX2 + BMI2, chosen at run time): the characters are folded to a letter index, looked up in a 26-entry table
//     with two byte shuffles, and -- if every one of the 32 is a plain state -- squeezed with bit-plane moves (DNA: two
//     vpmovmskb + two pdep) or four pext (amino acids);
//   * a block holding anything else (an ambiguity code, an unsupported character, a letter the two cases of which the alphabet
//     treats differently) takes the table-driven scalar path, which is also the portable fallback and the definition.
#include "rk_pack_host.h"

#include <string.h>

#if defined(__x86_64__)
#include <immintrin.h>
#define RK_PACK_X86 1
#else
#define RK_PACK_X86 0
#endif

namespace rk {

namespace {

// one block of n <= 32 symbols -> `nw` words at `dst` (the definition; every other path must equal it)
inline uint32_t pack_block_scalar(const unsigned char *table, uint32_t bits, const uint8_t *s, uint32_t n, uint32_t *dst, uint32_t nw) {
    uint32_t fl = 0, w = 0, have = 0;
    uint64_t acc = 0;
    for (uint32_t i = 0; i < n; i++) {
        const uint32_t c = table[s[i]];
        uint32_t st = c;
        if (c == 0xFF) { fl |= RK_FLAG_BAD_CHAR; st = 0; }
        else if (c & 0x80) { fl |= RK_FLAG_AMBIGUOUS; st = 0; }
        acc |= (uint64_t)st << have;
        have += bits;
        if (have >= 32) { if (w < nw) dst[w] = (uint32_t)acc; w++; acc >>= 32; have -= 32; }
    }
    if (w < nw) dst[w++] = (uint32_t)acc;
    while (w < nw) dst[w++] = 0;
    return fl;
}

#if RK_PACK_X86
struct LetterLut {
    alignas(32) unsigned char lo[32];  // letters a..p, twice (one copy per 128-bit lane of the shuffle)
    alignas(32) unsigned char hi[32];  // letters q..z + six invalid slots, twice
};

// 32 characters -> 32 states (0..31), or a byte >= 0x80 wherever the block needs the scalar path
__attribute__((target("avx2"))) inline __m256i classify32(const LetterLut &L, __m256i c) {
    const __m256i x = _mm256_or_si256(c, _mm256_set1_epi8(0x20));                     // fold the case
    const __m256i idx = _mm256_sub_epi8(x, _mm256_set1_epi8('a'));                     // a..z -> 0..25
    // a letter of either case: 0x41..0x5A / 0x61..0x7A  <=>  bits 7:6 of c are 01 and idx in 0..25
    const __m256i is_letter = _mm256_and_si256(_mm256_cmpeq_epi8(_mm256_and_si256(c, _mm256_set1_epi8((char)0xC0)), _mm256_set1_epi8(0x40)),
                                               _mm256_and_si256(_mm256_cmpgt_epi8(_mm256_set1_epi8(26), idx), _mm256_cmpgt_epi8(idx, _mm256_set1_epi8(-1))));
    const __m256i lo = _mm256_shuffle_epi8(_mm256_load_si256((const __m256i *)L.lo), idx);  // (bit 7 of idx is clear for letters)
    const __m256i hi = _mm256_shuffle_epi8(_mm256_load_si256((const __m256i *)L.hi), idx);
    const __m256i st = _mm256_blendv_epi8(lo, hi, _mm256_slli_epi16(idx, 3));         // bit 4 of idx -> bit 7: the upper half of the table
    return _mm256_or_si256(st, _mm256_andnot_si256(is_letter, _mm256_set1_epi8((char)0xFF)));
}

__attribute__((target("avx2,bmi2"))) inline void squeeze32_dna(__m256i st, uint32_t *dst, uint32_t nw) {
    // bit planes: plane b of the 32 states is one vpmovmskb; the two planes interleave into 64 bits
    const uint32_t p0 = (uint32_t)_mm256_movemask_epi8(_mm256_slli_epi16(st, 7));
    const uint32_t p1 = (uint32_t)_mm256_movemask_epi8(_mm256_slli_epi16(st, 6));
    const uint64_t v = _pdep_u64(p0, 0x5555555555555555ull) | _pdep_u64(p1, 0xAAAAAAAAAAAAAAAAull);
    if (nw >= 2) memcpy(dst, &v, 8);
    else if (nw == 1) dst[0] = (uint32_t)v;
}

__attribute__((target("avx2,bmi2"))) inline void squeeze32_aa(__m256i st, uint32_t *dst, uint32_t nw) {
    alignas(32) uint64_t q[4];
    _mm256_store_si256((__m256i *)q, st);
    const uint64_t m = 0x1F1F1F1F1F1F1F1Full;
    const uint64_t v0 = _pext_u64(q[0], m), v1 = _pext_u64(q[1], m), v2 = _pext_u64(q[2], m), v3 = _pext_u64(q[3], m);  // 40 bits each
    uint32_t out[5];
    out[0] = (uint32_t)v0;
    out[1] = (uint32_t)(v0 >> 32) | (uint32_t)(v1 << 8);
    out[2] = (uint32_t)(v1 >> 24) | (uint32_t)(v2 << 16);
    out[3] = (uint32_t)(v2 >> 16) | (uint32_t)(v3 << 24);
    out[4] = (uint32_t)(v3 >> 8);
    if (nw >= 5) memcpy(dst, out, 20);
    else memcpy(dst, out, 4 * nw);
}

template <int BITS>
__attribute__((target("avx2,bmi2"))) uint32_t pack_range_avx2(const PackSpec &P, const LetterLut &L, const uint8_t *seq, const uint64_t *off, uint64_t lo,
                                                              uint64_t hi, uint64_t r_base, uint32_t *packed, uint32_t *lens, uint32_t *flags) {
    uint32_t any = 0;
    constexpr uint32_t WPB = BITS == 2 ? 2 : 5;  // words per block of 32 symbols
    const uint32_t wpr = P.words_per_read, cap_syms = (wpr * 32u) / BITS;
    alignas(32) uint8_t tail[32];
    for (uint64_t r = lo; r < hi; r++) {
        const uint64_t o0 = off[r], full = off[r + 1] - o0;
        const uint32_t R = full > cap_syms ? cap_syms : (uint32_t)full;
        uint32_t fl = (full > cap_syms ? RK_FLAG_TOO_LONG : 0u) | (R < P.k ? RK_FLAG_TOO_SHORT : 0u);
        uint32_t *rec = packed + (r - r_base) * wpr;
        const uint8_t *s = seq + o0;
        uint32_t i = 0, w = 0;
        for (; i + 32 <= R; i += 32, w += WPB) {
            const __m256i st = classify32(L, _mm256_loadu_si256((const __m256i *)(s + i)));
            const uint32_t nw = wpr - w < WPB ? wpr - w : WPB;
            if (__builtin_expect(_mm256_movemask_epi8(st) != 0, 0)) fl |= pack_block_scalar(P.table, BITS, s + i, 32, rec + w, nw);
            else if (BITS == 2) squeeze32_dna(st, rec + w, nw);
            else squeeze32_aa(st, rec + w, nw);
        }
        if (i < R) {  // the last, partial block: padded with a letter of state 0, so the bits beyond the read are zero
            const uint32_t n = R - i;
            memset(tail, P.pad_char, 32);
            memcpy(tail, s + i, n);
            const __m256i st = classify32(L, _mm256_load_si256((const __m256i *)tail));
            const uint32_t nw = wpr - w < WPB ? wpr - w : WPB;
            if (__builtin_expect(_mm256_movemask_epi8(st) != 0, 0)) fl |= pack_block_scalar(P.table, BITS, s + i, n, rec + w, nw);
            else if (BITS == 2) squeeze32_dna(st, rec + w, nw);
            else squeeze32_aa(st, rec + w, nw);
            w += nw;
        }
        if (w > wpr) w = wpr;
        while (w < wpr) rec[w++] = 0;
        lens[r - r_base] = R;
        flags[r - r_base] = fl;
        any |= fl;
    }
    return any;
}

bool build_lut(const PackSpec &P, LetterLut &L) {
    for (int i = 0; i < 32; i++) {
        unsigned char v = 0xFF;
        if (i < 26) {
            const unsigned char up = P.table[(unsigned char)('A' + i)], lw = P.table[(unsigned char)('a' + i)];
            if (up == lw && up < 0x80) v = up;  // a plain state, the same for both cases; anything else takes the scalar path
        }
        if (i < 16) L.lo[i] = L.lo[i + 16] = v; else L.hi[i - 16] = L.hi[i] = v;
    }
    return P.table[P.pad_char] == 0 && __builtin_cpu_supports("avx2") && __builtin_cpu_supports("bmi2");
}
#endif  // RK_PACK_X86

uint32_t pack_range_scalar(const PackSpec &P, const uint8_t *seq, const uint64_t *off, uint64_t lo, uint64_t hi, uint64_t r_base, uint32_t *packed,
                           uint32_t *lens, uint32_t *flags) {
    uint32_t any = 0;
    const uint32_t wpr = P.words_per_read, cap_syms = (wpr * 32u) / P.bits, wpb = P.bits == 2 ? 2 : 5;
    for (uint64_t r = lo; r < hi; r++) {
        const uint64_t o0 = off[r], full = off[r + 1] - o0;
        const uint32_t R = full > cap_syms ? cap_syms : (uint32_t)full;
        uint32_t fl = (full > cap_syms ? RK_FLAG_TOO_LONG : 0u) | (R < P.k ? RK_FLAG_TOO_SHORT : 0u);
        uint32_t *rec = packed + (r - r_base) * wpr;
        uint32_t w = 0;
        for (uint32_t i = 0; i < R; i += 32, w += wpb) {
            const uint32_t n = R - i < 32 ? R - i : 32, nw = wpr - w < wpb ? wpr - w : wpb;
            fl |= pack_block_scalar(P.table, P.bits, seq + o0 + i, n, rec + w, nw);
        }
        if (w > wpr) w = wpr;
        while (w < wpr) rec[w++] = 0;
        lens[r - r_base] = R;
        flags[r - r_base] = fl;
        any |= fl;
    }
    return any;
}

}  // namespace

bool pack_reads_vectorised(const PackSpec &P) {
#if RK_PACK_X86
    LetterLut L;
    return !P.force_scalar && (P.bits == 2 || P.bits == 5) && build_lut(P, L);
#else
    (void)P;
    return false;
#endif
}

uint32_t pack_reads_range(const PackSpec &P, const uint8_t *seq, const uint64_t *off, uint64_t lo, uint64_t hi, uint64_t r_base, uint32_t *packed,
                          uint32_t *lens, uint32_t *flags) {
#if RK_PACK_X86
    LetterLut L;
    if (!P.force_scalar && (P.bits == 2 || P.bits == 5) && build_lut(P, L))
        return P.bits == 2 ? pack_range_avx2<2>(P, L, seq, off, lo, hi, r_base, packed, lens, flags)
                           : pack_range_avx2<5>(P, L, seq, off, lo, hi, r_base, packed, lens, flags);
#endif
    return pack_range_scalar(P, seq, off, lo, hi, r_base, packed, lens, flags);
}

}  // namespace rk
