// Developer repro of a gfx950 code-generation hazard found in round 2: a 5-bit packer that accumulates in 64 bits
// (variants 1 and 2: `acc |= (u64)st >> (-shift)`) loses the straddling symbol's bits in workgroups after the first 256,
// non-deterministically; the 32-bit formulation (variants 3, 4 and the product's pack_reads_kernel<5>) is exact.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -o pack_repro scripts/ubench/pack_repro.hip && ./pack_repro
#include "../../rappas_amd/csrc/rk_kernels.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstring>
using namespace rk;

// ---- candidate formulations (VAR: 1 = dword LDS table, 2 = table read from global memory, 3 = 32-bit shifts, 4 = 2 + 3) ----
template <int BITS, int VAR>
__global__ void __launch_bounds__(256) pack_variant(const unsigned char *ascii, const u64 *seq_off, u64 n_reads, u32 words_per_read,
                                                    const unsigned char *char_table, u32 k, u32 *packed, u32 *lens, u32 *flags) {
    __shared__ u32 tab32[256];
    tab32[threadIdx.x & 255] = char_table[threadIdx.x & 255];
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
        if (w == 0) { lens[r] = R; if (Rfull > cap_syms) fl |= RK_FLAG_TOO_LONG; if (R < k) fl |= RK_FLAG_TOO_SHORT; }
        const u32 s_lo = (32u * w) / BITS;
        const u32 s_hi = (32u * w + 31u) / BITS;
        u64 acc = 0;
        u32 acc32 = 0;
        for (u32 sidx = s_lo; sidx <= s_hi; sidx++) {
            if (sidx >= R) break;
            const u32 ch = ascii[o0 + sidx];
            u32 c = (VAR == 2 || VAR == 4) ? (u32)char_table[ch] : tab32[ch];
            u32 st = c;
            if (c == 0xFF) { fl |= RK_FLAG_BAD_CHAR; st = 0; }
            else if (c & 0x80) { fl |= RK_FLAG_AMBIGUOUS; st = 0; }
            if (VAR >= 3) {
                const u32 lo = sidx * BITS, base = 32u * w;
                acc32 |= lo >= base ? st << (lo - base) : st >> (base - lo);
            } else {
                int shift = (int)(sidx * BITS) - (int)(32u * w);
                if (shift >= 0) acc |= (u64)st << shift; else acc |= (u64)st >> (-shift);
            }
        }
        packed[t] = VAR >= 3 ? acc32 : (u32)acc;
        if (fl) atomicOr(&flags[r], fl);
    }
}

int main() {
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
    for (u64 r = 0; r < n; r++) {
        unsigned __int128 acc = 0; (void)acc;
        for (u64 i = 0; i < off[r + 1] - off[r]; i++) {
            u32 c = table[seq[off[r] + i]], st = (c & 0x80) ? 0 : c;
            u64 bit = i * 5;
            want[r * wpr + bit / 32] |= st << (bit % 32);
            if (bit % 32 > 27) want[r * wpr + bit / 32 + 1] |= st >> (32 - bit % 32);
        }
    }
    unsigned char *d_seq, *d_tab; u64 *d_off; u32 *d_packed, *d_lens, *d_flags;
    hipMalloc(&d_seq, seq.size()); hipMalloc(&d_tab, 256); hipMalloc(&d_off, (n + 1) * 8);
    hipMalloc(&d_packed, n * wpr * 4); hipMalloc(&d_lens, n * 4); hipMalloc(&d_flags, n * 4);
    hipMemcpy(d_seq, seq.data(), seq.size(), hipMemcpyHostToDevice);
    hipMemcpy(d_tab, table.data(), 256, hipMemcpyHostToDevice);
    hipMemcpy(d_off, off.data(), (n + 1) * 8, hipMemcpyHostToDevice);
    std::vector<u32> got(n * wpr);
    for (int var = 1; var <= 4; var++)
        for (int trial = 0; trial < 2; trial++) {
            hipMemset(d_flags, 0, n * 4);
            hipMemset(d_packed, 0xEE, n * wpr * 4);
            const u64 total = n * wpr;
            unsigned blocks = (unsigned)((total + 255) / 256);
            if (var == 1) hipLaunchKernelGGL((pack_variant<5, 1>), dim3(blocks), dim3(256), 0, 0, d_seq, (const u64 *)d_off, n, wpr, d_tab, k, d_packed, d_lens, d_flags);
            if (var == 2) hipLaunchKernelGGL((pack_variant<5, 2>), dim3(blocks), dim3(256), 0, 0, d_seq, (const u64 *)d_off, n, wpr, d_tab, k, d_packed, d_lens, d_flags);
            if (var == 3) hipLaunchKernelGGL((pack_variant<5, 3>), dim3(blocks), dim3(256), 0, 0, d_seq, (const u64 *)d_off, n, wpr, d_tab, k, d_packed, d_lens, d_flags);
            if (var == 4) hipLaunchKernelGGL((pack_variant<5, 4>), dim3(blocks), dim3(256), 0, 0, d_seq, (const u64 *)d_off, n, wpr, d_tab, k, d_packed, d_lens, d_flags);
            hipDeviceSynchronize();
            hipMemcpy(got.data(), d_packed, n * wpr * 4, hipMemcpyDeviceToHost);
            u64 bad = 0;
            for (u64 i = 0; i < n * wpr; i++) bad += got[i] != want[i];
            printf("variant %d trial %d: %llu differing words\n", var, trial, bad);
        }
    for (int trial = 0; trial < 4; trial++) {
        hipMemset(d_flags, 0, n * 4);
        hipMemset(d_packed, 0xEE, n * wpr * 4);
        const u64 total = n * wpr;
        unsigned blocks = (unsigned)((total + 255) / 256);
        if (trial >= 2 && blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(pack_reads_kernel<5>, dim3(blocks), dim3(256), 0, 0, d_seq, (const u64 *)d_off, n, wpr, d_tab, k, d_packed, d_lens, d_flags);
        hipDeviceSynchronize();
        hipMemcpy(got.data(), d_packed, n * wpr * 4, hipMemcpyDeviceToHost);
        u64 bad = 0, first = ~0ull;
        for (u64 i = 0; i < n * wpr; i++) if (got[i] != want[i]) { bad++; if (first == ~0ull) first = i; }
        printf("trial %d blocks %u: %llu differing words, first at row %llu word %llu (got %08x want %08x)\n", trial, blocks, bad,
               first / wpr, first % wpr, first == ~0ull ? 0 : got[first], first == ~0ull ? 0 : want[first]);
    }
    return 0;
}
