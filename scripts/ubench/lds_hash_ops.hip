// Micro-benchmark (round 4): what the LDS operations of a hash accumulator cost on gfx950 -- returning compare-and-swap, returning
// integer max / add, plain read -- as THROUGHPUT (independent operations, W waves per CU) and as LATENCY (every operation's address
// depends on the previous result), lane i -> word i ("distinct") or a pseudo-random word of 2 048 ("random": bank conflicts as in a
// hash table).  Diagnostic only.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32;
template <int OP, bool CHAIN, bool RANDOM>
__global__ void __launch_bounds__(64) k(u32 *out, int iters) {
    extern __shared__ u32 lds[];
    const u32 lane = threadIdx.x;
    for (u32 i = lane; i < 2048; i += 64) lds[i] = 0;
    __syncthreads();
    u32 idx = RANDOM ? ((lane * 0x9E3779B1u) >> 21) : lane;
    u32 acc = 0;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            u32 *q = lds + idx;
            u32 r;
            if (OP == 0) r = *(volatile u32 *)q;
            if (OP == 1) r = atomicCAS(q, 0u, lane + 1u);
            if (OP == 2) r = atomicMax(q, (u32)i);
            if (OP == 3) r = atomicAdd(q, 1u);
            if (OP == 4) r = atomicExch(q, (u32)i);
            if (OP == 5) { r = *(volatile u32 *)q; *(volatile u32 *)q = r + 1u; }
            acc += r;
            if (CHAIN) idx = (idx + (r & 1u) * 64u + 64u) & 2047u;  // the next address needs this result
            else idx = (idx + 64u) & 2047u;
        }
    }
    if (lane == 0) out[blockIdx.x] = acc;
}
template <int OP, bool CHAIN, bool RANDOM>
void run(const char *name, int waves) {
    const int blocks = 256 * waves, iters = 2000;
    u32 *d;
    hipMalloc(&d, blocks * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const size_t lds = waves == 8 ? 20480 : 65536 * 2;  // (one block per CU when waves == 1)
    hipFuncSetAttribute((const void *)k<OP, CHAIN, RANDOM>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    k<OP, CHAIN, RANDOM><<<blocks, 64, lds>>>(d, 10);
    hipEventRecord(e0);
    k<OP, CHAIN, RANDOM><<<blocks, 64, lds>>>(d, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double per_wave = ms * 1e6 / (iters * 8.0);  // ns per operation of one wave
    printf("%-28s %-9s %-8s %d wave%s/CU  %8.3f ms  %7.1f ns = ~%6.1f cycles per operation of a wave (2.4 GHz)%s\n", name, CHAIN ? "chained" : "indep.", RANDOM ? "random" : "distinct", waves,
           waves == 1 ? " " : "s", ms, per_wave, per_wave * 2.4, CHAIN ? "" : (waves == 8 ? "  [/8 = per CU]" : ""));
    hipFree(d);
}
template <int OP>
void all(const char *name) {
    run<OP, true, false>(name, 1);
    run<OP, true, true>(name, 1);
    run<OP, true, true>(name, 8);
    run<OP, false, false>(name, 8);
    run<OP, false, true>(name, 8);
}
int main() {
    all<0>("ds_read_b32");
    all<1>("ds_cmpst_rtn_b32");
    all<2>("ds_max_rtn_u32");
    all<3>("ds_add_rtn_u32");
    all<4>("ds_wrxchg_rtn_b32");
    all<5>("ds_read_b32 + ds_write_b32");
    return 0;
}
