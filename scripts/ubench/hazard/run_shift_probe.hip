// Runs the 64-bit shift count-operand probes (make_shift_probe.py): per code object, how many lanes ever got a different result
// from v_lshrrev_b64 / v_lshlrev_b64 / v_ashrrev_i64 when the 32-bit count sits in the LAST allocated VGPR, and what came out.
//   hipcc -O2 -o run_shift_probe run_shift_probe.hip && ./run_shift_probe shift_probe_24.co ...
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
typedef unsigned int u32;
typedef unsigned long long u64;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

int main(int argc, char **argv) {
    const unsigned blocks = 8192, rounds = 400;
    const size_t n_thr = (size_t)blocks * 256;
    u32 *d_out;
    CK(hipMalloc(&d_out, n_thr * 32));
    std::vector<u32> out(n_thr * 8);
    for (int ai = 1; ai < argc; ai++) {
        hipModule_t mod; hipFunction_t fn;
        CK(hipModuleLoad(&mod, argv[ai]));
        CK(hipModuleGetFunction(&fn, mod, "shift_probe"));
        for (int trial = 0; trial < 2; trial++) {
            CK(hipMemset(d_out, 0xCD, n_thr * 32));
            u32 *a0 = d_out; u32 r = rounds;
            void *args[] = {&a0, &r};
            CK(hipModuleLaunchKernel(fn, blocks, 1, 1, 256, 1, 1, 0, 0, args, nullptr));
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(out.data(), d_out, n_thr * 32, hipMemcpyDeviceToHost));
            size_t wrong = 0, missing = 0, waves = 0, every = 0; std::map<u32, size_t> first_round, times; int shown = 0;
            for (size_t t = 0; t < n_thr; t++) {
                const u32 *rec = &out[t * 8];
                if (rec[0] != (u32)t) { missing++; continue; }
                if (!rec[1]) continue;
                wrong++;
                if ((t & 63) == 0) waves++;          // (whole waves are affected together: lane 0 stands for its wave)
                every += rec[1] == rounds;
                first_round[rec[2] - 1 < 4 ? rec[2] - 1 : (rec[2] - 1 < 32 ? 4 : (rec[2] - 1 < 200 ? 32 : 200))]++;
                times[rec[1] < 4 ? rec[1] : 4]++;
                if (shown < 3 && (t & 63) < 3) { shown++; printf("    thread %zu (lane %zu): wrong in %u of %u rounds, first in round %u; wrong result %08x:%08x, value.lo %08x\n",
                                                 t, t & 63, rec[1], rounds, rec[2] - 1, rec[5], rec[4], rec[6]); }
            }
            printf("%-44s trial %d: %8zu of %zu lanes (%zu waves) got a wrong result at least once; in every round: %zu; times wrong (1,2,3,4+):", argv[ai], trial, wrong, n_thr, waves, every);
            for (u32 k = 1; k <= 4; k++) printf(" %zu", times.count(k) ? times[k] : 0);
            printf("; first wrong round (0,1,2,3,4-31,32-199,200+):");
            for (u32 k : {0u, 1u, 2u, 3u, 4u, 32u, 200u}) printf(" %zu", first_round.count(k) ? first_round[k] : 0);
            printf("%s\n", missing ? "  (records missing!)" : "");
        }
        CK(hipModuleUnload(mod));
    }
    return 0;
}
