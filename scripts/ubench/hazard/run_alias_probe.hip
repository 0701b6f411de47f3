// Runs the VGPR aliasing probes (make_alias_probe.py): for every code object given, launches `alias_probe` on a grid large
// enough that waves come and go on every SIMD while others are still checking, and reports how many threads found one of
// their registers overwritten, which registers, and what they held.
//   hipcc -O2 -o run_alias_probe run_alias_probe.hip && ./run_alias_probe a.co b.co ...
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <vector>
typedef unsigned int u32;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

int main(int argc, char **argv) {
    const unsigned blocks = 8192, rounds = 200;
    const size_t n_thr = (size_t)blocks * 256;
    u32 *d_out;
    CK(hipMalloc(&d_out, n_thr * 32));
    std::vector<u32> out(n_thr * 8);
    for (int ai = 1; ai < argc; ai++) {
        hipModule_t mod; hipFunction_t fn;
        CK(hipModuleLoad(&mod, argv[ai]));
        CK(hipModuleGetFunction(&fn, mod, "alias_probe"));
        for (int trial = 0; trial < 2; trial++) {
            CK(hipMemset(d_out, 0xCD, n_thr * 32));
            u32 *a0 = d_out; u32 r = rounds;
            void *args[] = {&a0, &r};
            CK(hipModuleLaunchKernel(fn, blocks, 1, 1, 256, 1, 1, 0, 0, args, nullptr));
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(out.data(), d_out, n_thr * 32, hipMemcpyDeviceToHost));
            size_t bad_thr = 0, bad_waves = 0, missing = 0; u32 regmask = 0; std::map<u32, size_t> first_reg; int shown = 0;
            for (size_t t = 0; t < n_thr; t++) {
                const u32 *rec = &out[t * 8];
                if (rec[0] != (u32)t) { missing++; continue; }
                if ((t & 63) == 0 && rec[1]) bad_waves++;
                const unsigned long long lanes = rec[5] | ((unsigned long long)rec[6] << 32);
                if (!((lanes >> (t & 63)) & 1)) continue;
                bad_thr++;
                regmask |= rec[1];
                first_reg[rec[2] - 1]++;
                if (shown < 4) { shown++; printf("    thread %zu (workgroup %zu wave %zu lane %zu): registers-wrong mask %08x, first wrong v%u in round %u, top register now holds expected ^ %08x\n",
                                                 t, t >> 8, (t >> 6) & 3, t & 63, rec[1], rec[2] - 1, rec[3] - 1, rec[4]); }
            }
            printf("%-34s trial %d: %zu of %zu threads saw a register of theirs change (%zu waves); records missing %zu; registers hit (bit = index mod 32): %08x; first-hit register histogram:",
                   argv[ai], trial, bad_thr, n_thr, bad_waves, missing, regmask);
            for (auto &kv : first_reg) printf(" v%u:%zu", kv.first, kv.second);
            printf("\n");
        }
        CK(hipModuleUnload(mod));
    }
    return 0;
}
