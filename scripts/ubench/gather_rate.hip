// Micro-benchmark: random-gather rate from a cache-resident table (Infinity Cache / L2) on MI355X, with the access
// shape of the placement kernel's row loads (groups of GW lanes read GW consecutive 8-byte words of a random,
// ALIGN-aligned row; U loads in flight per lane; W waves per CU).  Address generation is a xorshift + mask so that
// the loop is not VALU-bound.  Output: one line per configuration.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned long long u64;
typedef unsigned int u32;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s\n", hipGetErrorString(e_)); return 1; } } while (0)

template <int GW, int U>
__global__ void __launch_bounds__(64) gather(const u64 *tab, u32 row_mask, u32 row_words, u32 iters, u64 *out) {
    const u32 lane = threadIdx.x & 63, li = lane % GW, gi = lane / GW;
    u64 acc = 0;
    u32 x = (blockIdx.x * 4 + gi) * 2654435761u + 12345u;
    for (u32 it = 0; it < iters; it += U) {
        u64 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            x ^= x << 13; x ^= x >> 17; x ^= x << 5;
            v[u] = tab[(size_t)(x & row_mask) * row_words + li];
        }
#pragma unroll
        for (int u = 0; u < U; u++) acc += v[u];
    }
    if (acc == 0x1234567) out[0] = acc;
}

template <int GW, int U>
static int run(const u64 *d, u64 n_words, u32 row_words, int waves_per_cu, u64 *dout) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    u32 rows = 1; while ((u64)rows * 2 * row_words <= n_words) rows *= 2;
    const u32 iters = 8192;
    const int blocks = 256 * waves_per_cu;
    gather<GW, U><<<blocks, 64>>>(d, rows - 1, row_words, 64, dout);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    gather<GW, U><<<blocks, 64>>>(d, rows - 1, row_words, iters, dout);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double group_loads = (double)blocks * (64 / GW) * iters;
    printf("table=%6.1f MB  group=%2d lanes (%3d B rows, stride %4u B)  U=%2d  waves/CU=%2d : %7.2f ms  %.3e rows/s  %.2f TB/s useful  %.3e wave-instr/s/CU\n",
           (double)rows * row_words * 8 / 1e6, GW, GW * 8, row_words * 8, U, waves_per_cu, ms, group_loads / (ms * 1e-3),
           group_loads * GW * 8 / (ms * 1e-3) / 1e12, (double)blocks * iters / (ms * 1e-3) / 256);
    return 0;
}

int main() {
    const u64 n_words = (u64)(256e6 / 8);
    u64 *d, *dout;
    CK(hipMalloc(&d, n_words * 8)); CK(hipMalloc(&dout, 8));
    CK(hipMemset(d, 1, n_words * 8));
    for (u64 words : {(u64)(2e6 / 8), (u64)(100e6 / 8)}) {
        for (int w : {8, 16, 32}) {
            if (run<1, 8>(d, words, 8, w, dout)) return 1;     // lane-random 8 B (table probes), 64-byte stride
            if (run<8, 8>(d, words, 8, w, dout)) return 1;     // 64-byte rows
            if (run<16, 8>(d, words, 16, w, dout)) return 1;   // 128-byte rows (the kernel's chunks)
            if (run<16, 16>(d, words, 16, w, dout)) return 1;
            if (run<64, 8>(d, words, 64, w, dout)) return 1;   // 512-byte rows
        }
    }
    return 0;
}
