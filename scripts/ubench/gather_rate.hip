// Micro-benchmark: random-gather request rate from a cache-resident table (Infinity Cache / L2) on MI355X.
// Calibrates the ceiling the placement kernel's row / table gathers can reach (DESIGN.md "Roofline").
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned long long u64;
typedef unsigned int u32;

__device__ __forceinline__ u64 mix(u64 x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }

// every group of GW lanes reads GW consecutive 8-byte words starting at a random 8-byte-aligned (or ALIGN-aligned) offset
template <int GW, int U>
__global__ void __launch_bounds__(64) gather(const u64 *tab, u64 n_words, u32 iters, u32 align_words, u64 *out) {
    const u32 lane = threadIdx.x & 63, li = lane % GW, gi = lane / GW;
    u64 acc = 0;
    u64 seed = ((u64)blockIdx.x * 64 + gi) * 0x9E3779B97F4A7C15ULL + 12345;
    for (u32 it = 0; it < iters; it += U) {
        u64 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            u64 r = mix(seed + (u64)(it + u) * 977);
            u64 base = (r % (n_words / align_words - 2)) * align_words;
            v[u] = tab[base + li];
        }
#pragma unroll
        for (int u = 0; u < U; u++) acc += v[u];
    }
    if (acc == 0x1234567) out[0] = acc;
}

template <int GW, int U>
static void run(const u64 *d, u64 n_words, u32 align_words, u64 *dout, const char *name) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const u32 iters = 4096;
    const int blocks = 256 * 8;
    gather<GW, U><<<blocks, 64>>>(d, n_words, 64, align_words, dout);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    gather<GW, U><<<blocks, 64>>>(d, n_words, iters, align_words, dout);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double group_loads = (double)blocks * (64 / GW) * iters;
    double lines = group_loads * ((GW * 8 + 63) / 64 + (align_words * 8 % 64 ? 0.5 : 0));
    printf("%-28s table=%6.1f MB group=%2d lanes (%4d B) align=%3u B U=%2d : %.2f ms  %.3e group-gathers/s  ~%.3e 64B-lines/s  %.2f TB/s useful\n",
           name, n_words * 8 / 1e6, GW, GW * 8, align_words * 8, U, ms, group_loads / (ms * 1e-3), lines / (ms * 1e-3),
           group_loads * GW * 8 / (ms * 1e-3) / 1e12);
}

int main() {
    for (double mb : {2.0, 8.0, 80.0, 160.0}) {
        u64 n_words = (u64)(mb * 1e6 / 8);
        u64 *d, *dout;
        hipMalloc(&d, n_words * 8); hipMalloc(&dout, 8);
        hipMemset(d, 1, n_words * 8);
        run<1, 8>(d, n_words, 1, dout, "lane-random 8B");
        run<8, 8>(d, n_words, 1, dout, "8-lane rows unaligned");
        run<8, 8>(d, n_words, 8, dout, "8-lane rows 64B-aligned");
        run<16, 8>(d, n_words, 1, dout, "16-lane rows unaligned");
        run<16, 8>(d, n_words, 16, dout, "16-lane rows 128B-aligned");
        run<16, 16>(d, n_words, 1, dout, "16-lane rows unaligned U16");
        run<64, 8>(d, n_words, 64, dout, "64-lane rows 512B-aligned");
        hipFree(d); hipFree(dout);
    }
    return 0;
}
