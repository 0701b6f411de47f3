// Root-cause bisect of the round-2 "5-bit packer loses the straddling symbol" corruption.
// Same data and the same failing formulation as scripts/ubench/pack_repro.hip (variant 1: 64-bit accumulate,
// `acc |= (u64)st >> (-shift)`), plus controlled variations and a register-file poison kernel:
//   A  the failing formulation as it was
//   B  A on a grid of 256 workgroups (one per CU, grid-stride loop): no second workgroup ever shares a CU
//   C  A with s_nop 7 around the shifts (a missing wait state would be covered)
//   D  A's loop and control flow with 32-bit shifts (is it the loop or the shift?)
//   E  A with branch-free symbol classification (no if/else around the shifted value: no implicit-def)
//   F  A with the early `break` replaced by a predicated body
//   G  A with the left / right select done by a sign mask (no v_cmp -> vcc -> v_cndmask behind the 64-bit shifts)
//   H  A with the compare hoisted in front of the shifts into an SGPR pair (ballot), select by that pair
// Every variant runs after (i) nothing, (ii) a kernel that fills the whole VGPR/AGPR file of every SIMD with ones,
// (iii) the same with zeros.  A result that follows the poison value is an uninitialised-register read.
// For the failing words the wave's HW_ID / XCC_ID, lane and word index inside the read are tallied.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o pack_hazard scripts/ubench/pack_hazard.hip && ./pack_hazard
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <vector>
typedef unsigned long long u64;
typedef unsigned int u32;

#define FLAG_BAD 2u
#define FLAG_AMB 8u

template <int VAR>
__global__ void __launch_bounds__(256) pack_variant(const unsigned char *ascii, const u64 *seq_off, u64 n_reads, u32 words_per_read,
                                                    const unsigned char *char_table, u32 k, u32 *packed, u32 *lens, u32 *flags, u32 *hwid) {
    constexpr int BITS = 5;
    __shared__ u32 tab32[256];
    tab32[threadIdx.x & 255] = char_table[threadIdx.x & 255];
    __syncthreads();
    const u64 total = n_reads * words_per_read;
    for (u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (u64)gridDim.x * blockDim.x) {
        if ((t & 63) == 0) {
            hwid[2 * (t >> 6)] = __builtin_amdgcn_s_getreg((31 << 11) | 4);       // HW_ID
            hwid[2 * (t >> 6) + 1] = __builtin_amdgcn_s_getreg((31 << 11) | 20);  // XCC_ID
        }
        const u64 r = t / words_per_read;
        const u32 w = (u32)(t - r * words_per_read);
        const u64 o0 = seq_off[r];
        const u64 Rfull = seq_off[r + 1] - o0;
        const u32 cap_syms = (words_per_read * 32u) / BITS;
        const u32 R = Rfull > cap_syms ? cap_syms : (u32)Rfull;
        u32 fl = 0;
        if (w == 0) { lens[r] = R; if (R < k) fl |= 4u; }
        const u32 s_lo = (32u * w) / BITS;
        const u32 s_hi = (32u * w + 31u) / BITS;
        u64 acc = 0;
        u32 acc32 = 0;
        for (u32 sidx = s_lo; sidx <= s_hi; sidx++) {
            if (VAR != 'F') { if (sidx >= R) break; }
            if (VAR == 'F' && sidx >= R) continue;
            const u32 ch = ascii[o0 + sidx];
            u32 c = tab32[ch];
            u32 st = c;
            if (VAR == 'E') {
                const bool bad = c == 0xFF, amb = !bad && (c & 0x80);
                fl |= (bad ? FLAG_BAD : 0u) | (amb ? FLAG_AMB : 0u);
                st = (bad || amb) ? 0u : c;
            } else {
                if (c == 0xFF) { fl |= FLAG_BAD; st = 0; }
                else if (c & 0x80) { fl |= FLAG_AMB; st = 0; }
            }
            if (VAR == 'D') {
                const u32 lo = sidx * BITS, base = 32u * w;
                asm volatile("" : "+v"(st));  // keeps the loop scalar, like A's
                acc32 |= lo >= base ? st << (lo - base) : st >> (base - lo);
            } else {
                int shift = (int)(sidx * BITS) - (int)(32u * w);
                if (VAR == 'C') asm volatile("s_nop 7\n\ts_nop 7" : "+v"(shift), "+v"(st));
                if (VAR == 'G') {
                    const u64 l = (u64)st << (shift & 63), rr = (u64)st >> ((-shift) & 63), m = (u64)((long long)shift >> 63);
                    acc |= (rr & m) | (l & ~m);
                } else if (VAR == 'H') {
                    const u64 neg = __ballot(shift < 0);
                    const u64 l = (u64)st << (shift & 63), rr = (u64)st >> ((-shift) & 63);
                    acc |= ((neg >> (threadIdx.x & 63)) & 1) ? rr : l;
                } else if (shift >= 0) acc |= (u64)st << shift; else acc |= (u64)st >> (-shift);
                if (VAR == 'C') asm volatile("s_nop 7\n\ts_nop 7" : "+v"(acc));
            }
        }
        packed[t] = VAR == 'D' ? acc32 : (u32)acc;
        if (fl) atomicOr(&flags[r], fl);
    }
}

// fills v0..v255 and a0..a255 of the wave with `val`: 512 registers per lane = the whole file of its SIMD
__global__ void __launch_bounds__(256) poison_kernel(u32 val, u32 *sink) {
    u32 x = val;
#define P8(b) asm volatile("v_mov_b32 v" #b "0, %0\n\tv_mov_b32 v" #b "1, %0\n\tv_mov_b32 v" #b "2, %0\n\tv_mov_b32 v" #b "3, %0\n\tv_mov_b32 v" #b "4, %0\n\t" \
                           "v_mov_b32 v" #b "5, %0\n\tv_mov_b32 v" #b "6, %0\n\tv_mov_b32 v" #b "7, %0\n\tv_mov_b32 v" #b "8, %0\n\tv_mov_b32 v" #b "9, %0" :: "s"(x) \
                           : "v" #b "0", "v" #b "1", "v" #b "2", "v" #b "3", "v" #b "4", "v" #b "5", "v" #b "6", "v" #b "7", "v" #b "8", "v" #b "9");
    P8(1) P8(2) P8(3) P8(4) P8(5) P8(6) P8(7) P8(8) P8(9) P8(10) P8(11) P8(12) P8(13) P8(14) P8(15) P8(16) P8(17) P8(18) P8(19) P8(20) P8(21) P8(22) P8(23) P8(24)
    asm volatile("v_mov_b32 v250, %0\n\tv_mov_b32 v251, %0\n\tv_mov_b32 v252, %0\n\tv_mov_b32 v253, %0\n\tv_mov_b32 v254, %0\n\tv_mov_b32 v255, %0" :: "s"(x)
                 : "v250", "v251", "v252", "v253", "v254", "v255");
#define A8(b) asm volatile("v_accvgpr_write_b32 a" #b "0, %0\n\tv_accvgpr_write_b32 a" #b "1, %0\n\tv_accvgpr_write_b32 a" #b "2, %0\n\tv_accvgpr_write_b32 a" #b "3, %0\n\t" \
                           "v_accvgpr_write_b32 a" #b "4, %0\n\tv_accvgpr_write_b32 a" #b "5, %0\n\tv_accvgpr_write_b32 a" #b "6, %0\n\tv_accvgpr_write_b32 a" #b "7, %0\n\t" \
                           "v_accvgpr_write_b32 a" #b "8, %0\n\tv_accvgpr_write_b32 a" #b "9, %0" :: "s"(x) \
                           : "a" #b "0", "a" #b "1", "a" #b "2", "a" #b "3", "a" #b "4", "a" #b "5", "a" #b "6", "a" #b "7", "a" #b "8", "a" #b "9");
    A8(1) A8(2) A8(3) A8(4) A8(5) A8(6) A8(7) A8(8) A8(9) A8(10) A8(11) A8(12) A8(13) A8(14) A8(15) A8(16) A8(17) A8(18) A8(19) A8(20) A8(21) A8(22) A8(23) A8(24)
    asm volatile("v_accvgpr_write_b32 a0, %0\n\tv_accvgpr_write_b32 a1, %0\n\tv_accvgpr_write_b32 a2, %0\n\tv_accvgpr_write_b32 a3, %0\n\tv_accvgpr_write_b32 a4, %0\n\t"
                 "v_accvgpr_write_b32 a5, %0\n\tv_accvgpr_write_b32 a6, %0\n\tv_accvgpr_write_b32 a7, %0\n\tv_accvgpr_write_b32 a8, %0\n\tv_accvgpr_write_b32 a9, %0\n\t"
                 "v_accvgpr_write_b32 a250, %0\n\tv_accvgpr_write_b32 a251, %0\n\tv_accvgpr_write_b32 a252, %0\n\tv_accvgpr_write_b32 a253, %0\n\tv_accvgpr_write_b32 a254, %0\n\t"
                 "v_accvgpr_write_b32 a255, %0" :: "s"(x)
                 : "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a250", "a251", "a252", "a253", "a254", "a255");
    // low registers last, through the compiler's own allocation: ten live values written back so nothing is dead code
    u32 y[10];
#pragma unroll
    for (int i = 0; i < 10; i++) { y[i] = x; asm volatile("" : "+v"(y[i])); }
    u32 s = 0;
#pragma unroll
    for (int i = 0; i < 10; i++) s += y[i];
    if (s == 12345u) sink[threadIdx.x] = s;
}

static void launch(char var, unsigned blocks, const unsigned char *seq, const u64 *off, u64 n, u32 wpr, const unsigned char *tab, u32 k, u32 *packed, u32 *lens,
                   u32 *flags, u32 *hwid) {
#define L(V) case V: hipLaunchKernelGGL((pack_variant<V>), dim3(blocks), dim3(256), 0, 0, seq, off, n, wpr, tab, k, packed, lens, flags, hwid); break;
    switch (var) { L('A') L('C') L('D') L('E') L('F') L('G') L('H') }
}

int main(int argc, char **argv) {
    const u64 n = 50000;
    const u32 wpr = 19, k = 5;
    std::vector<unsigned char> table(256, 0xFF);
    const char *order = "RHKDESTNQCGPAILMFWYV";
    for (int i = 0; i < 20; i++) { table[(unsigned char)order[i]] = i; table[(unsigned char)order[i] + 32] = i; }
    table['-'] = 0x80;
    std::vector<u64> off(n + 1, 0);
    srand(7);
    for (u64 r = 0; r < n; r++) off[r + 1] = off[r] + 60 + rand() % 61;
    std::vector<unsigned char> seq(off[n]);
    for (auto &c : seq) c = (rand() % 200 == 0) ? '-' : order[rand() % 20];
    std::vector<u32> want(n * wpr, 0);
    for (u64 r = 0; r < n; r++)
        for (u64 i = 0; i < off[r + 1] - off[r]; i++) {
            u32 c = table[seq[off[r] + i]], st = (c & 0x80) ? 0 : c;
            u64 bit = i * 5;
            want[r * wpr + bit / 32] |= st << (bit % 32);
            if (bit % 32 > 27) want[r * wpr + bit / 32 + 1] |= st >> (32 - bit % 32);
        }
    unsigned char *d_seq, *d_tab; u64 *d_off; u32 *d_packed, *d_lens, *d_flags, *d_hwid, *d_sink;
    const u64 total = n * wpr;
    hipMalloc(&d_seq, seq.size()); hipMalloc(&d_tab, 256); hipMalloc(&d_off, (n + 1) * 8);
    hipMalloc(&d_packed, total * 4); hipMalloc(&d_lens, n * 4); hipMalloc(&d_flags, n * 4); hipMalloc(&d_hwid, (total / 64 + 2) * 8); hipMalloc(&d_sink, 4096);
    hipMemcpy(d_seq, seq.data(), seq.size(), hipMemcpyHostToDevice);
    hipMemcpy(d_tab, table.data(), 256, hipMemcpyHostToDevice);
    hipMemcpy(d_off, off.data(), (n + 1) * 8, hipMemcpyHostToDevice);
    std::vector<u32> got(total), hw((total / 64 + 2) * 2);
    const unsigned full = (unsigned)((total + 255) / 256);
    struct Run { char var; unsigned blocks; const char *name; };
    const Run runs[] = {{'A', full, "A 64-bit accumulate, as it failed"}, {'A', 256, "B = A on 256 workgroups (one per CU)"}, {'C', full, "C = A + s_nop around the shifts"},
                        {'D', full, "D = A's loop, 32-bit shifts"}, {'E', full, "E = A, branch-free classification"}, {'F', full, "F = A, predicated instead of break"},
                        {'G', full, "G = A, select by sign mask (no vcc)"}, {'H', full, "H = A, compare hoisted (ballot)"}};
    const char *poison_name[] = {"no poison", "registers poisoned with 0xFFFFFFFF", "registers poisoned with 0"};
    const int n_poison = argc > 1 ? atoi(argv[1]) : 3;
    for (int poison = 0; poison < n_poison; poison++)
        for (const Run &rn : runs)
            for (int trial = 0; trial < 2; trial++) {
                hipMemset(d_flags, 0, n * 4);
                hipMemset(d_packed, 0xEE, total * 4);
                hipMemset(d_hwid, 0, (total / 64 + 2) * 8);
                if (poison) { hipLaunchKernelGGL(poison_kernel, dim3(2048), dim3(256), 0, 0, poison == 1 ? 0xFFFFFFFFu : 0u, d_sink); }
                launch(rn.var, rn.blocks, d_seq, d_off, n, wpr, d_tab, k, d_packed, d_lens, d_flags, d_hwid);
                if (hipDeviceSynchronize() != hipSuccess) { printf("HIP error\n"); return 1; }
                hipMemcpy(got.data(), d_packed, total * 4, hipMemcpyDeviceToHost);
                hipMemcpy(hw.data(), d_hwid, (total / 64 + 2) * 8, hipMemcpyDeviceToHost);
                u64 bad = 0, bad_waves = 0, lost_only = 0, first = ~0ull;
                std::map<u32, u64> by_simd, by_waveslot, by_lane_half, by_wgpos, by_word;
                u64 last_wave = ~0ull;
                u64 extra_bits = 0;
                for (u64 i = 0; i < total; i++) {
                    if (got[i] == want[i]) continue;
                    bad++;
                    if (first == ~0ull) first = i;
                    const u32 missing = want[i] & ~got[i], extra = got[i] & ~want[i];
                    if (extra) extra_bits++;
                    if (!extra && (missing >> 5) == 0) lost_only++;   // only the low bits (the straddling symbol's upper part lands at bit 0..3) are missing
                    const u64 wv = i >> 6;
                    if (wv != last_wave) { bad_waves++; last_wave = wv; by_simd[(hw[2 * wv] >> 4) & 3]++; by_waveslot[hw[2 * wv] & 15]++; by_wgpos[(u32)((i >> 6) & 3)]++; }
                    by_lane_half[(u32)((i & 63) >> 5)]++;
                    by_word[(u32)(i % wpr)]++;
                }
                printf("%-42s | %-36s | trial %d: %7llu wrong words in %6llu waves (of %llu); only-straddle-bits-missing %llu, words with extra bits %llu", rn.name,
                       poison_name[poison], trial, bad, bad_waves, total / 64, lost_only, extra_bits);
                if (bad) {
                    printf("; first at word %llu (workgroup %llu) got %08x want %08x\n    waves by SIMD:", first, first / 256, got[first], want[first]);
                    for (auto &kv : by_simd) printf(" %u:%llu", kv.first, kv.second);
                    printf("  by wave slot:");
                    for (auto &kv : by_waveslot) printf(" %u:%llu", kv.first, kv.second);
                    printf("  by wave-in-workgroup:");
                    for (auto &kv : by_wgpos) printf(" %u:%llu", kv.first, kv.second);
                    printf("  words by lane half:");
                    for (auto &kv : by_lane_half) printf(" %u:%llu", kv.first, kv.second);
                    printf("  full waves wrong? %s", bad >= bad_waves * 20 ? "mostly" : "no, scattered lanes");
                }
                printf("\n");
            }
    return 0;
}
