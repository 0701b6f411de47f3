// Micro-benchmark: LDS operation rates on gfx950 with every CU's LDS pipe kept busy (8 waves per CU, 2 per SIMD): plain writes,
// non-returning float adds / integer ORs, read-modify-write by hand -- 64 or 16 active lanes, distinct addresses (stride 1 word) or
// one address per 16-lane group.  Diagnostic only (round 3: should the windowed stream use LDS atomics?).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u32;
template <int OP, int ACTIVE, bool SAME>
__global__ void __launch_bounds__(64) k(u32 *out, int iters) {
    extern __shared__ u32 lds[];
    const u32 lane = threadIdx.x;
    for (u32 i = lane; i < 2048; i += 64) lds[i] = 0;
    __syncthreads();
    u32 *p = lds + (SAME ? (lane & ~15u) : lane);
    float *pf = (float *)p;
    u32 acc = 0;
    const bool on = (lane % (64 / ACTIVE)) == 0;
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            u32 *q = p + 64 * j;
            if (on) {
                if (OP == 0) *(volatile u32 *)q = (u32)i;
                if (OP == 1) __hip_atomic_fetch_add((float *)q, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                if (OP == 2) __hip_atomic_fetch_or(q, 1u << (i & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                if (OP == 3) { const u32 o = *(volatile u32 *)q; *(volatile u32 *)q = o + 1; }
                if (OP == 4) acc += __hip_atomic_fetch_add(q, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                if (OP == 5) __hip_atomic_fetch_add(q, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                if (OP == 6) acc += __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);  // the SAME word, instruction after instruction
                if (OP == 7) __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            }
        }
    }
    __syncthreads();
    long long t1 = __builtin_readcyclecounter();
    if (lane == 0) { out[2 * blockIdx.x] = (u32)(t1 - t0); out[2 * blockIdx.x + 1] = acc + lds[lane]; }
    (void)pf;
}
template <int OP, int ACTIVE, bool SAME>
void run(const char *name) {
    const int blocks = 256 * 8, iters = 2000;
    u32 *d;
    hipMalloc(&d, blocks * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP, ACTIVE, SAME><<<blocks, 64, 20480>>>(d, 10);
    hipEventRecord(e0);
    k<OP, ACTIVE, SAME><<<blocks, 64, 20480>>>(d, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    // 8 waves a CU, each iters * 8 wave-instructions
    const double instr_per_cu = 8.0 * iters * 8;
    printf("%-34s active=%2d %-8s  %7.3f ms   %6.1f ns per wave-instruction per CU  (~%.1f cycles at 2.4 GHz)\n", name, ACTIVE, SAME ? "same/16" : "distinct", ms,
           ms * 1e6 / instr_per_cu, ms * 1e6 / instr_per_cu * 2.4);
    hipFree(d);
}
int main() {
    run<0, 64, false>("ds_write_b32");
    run<0, 16, false>("ds_write_b32");
    run<1, 64, false>("ds_add_f32 (no return)");
    run<1, 16, false>("ds_add_f32 (no return)");
    run<1, 64, true>("ds_add_f32 (no return)");
    run<2, 64, false>("ds_or_b32 (no return)");
    run<2, 16, false>("ds_or_b32 (no return)");
    run<2, 64, true>("ds_or_b32 (no return)");
    run<5, 64, false>("ds_add_u32 (no return)");
    run<4, 64, false>("ds_add_rtn_u32");
    run<6, 64, false>("ds_add_rtn_u32, same word again");
    run<6, 4, false>("ds_add_rtn_u32, same word again");
    run<7, 64, false>("ds_add_u32, same word again");
    run<3, 64, false>("ds_read_b32 + ds_write_b32");
    run<3, 16, false>("ds_read_b32 + ds_write_b32");
    return 0;
}
