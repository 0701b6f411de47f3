// Loads the instrumented build of the failing packer (pack_variant_A_instrumented.s, assembled to a code object) and reports,
// for every wrong output word, what the first inner-loop iteration (the one that handles the symbol straddling the word's lower
// edge) saw: the wave's vcc after `v_cmp_gt_i32 vcc, 0, shift`, the 64-bit left-shift result, the accumulator after the iteration
// and the shift registers (latched in registers, stored once behind the loop).
//   clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c pack_variant_A_instrumented.s -o a.o && ld.lld -shared a.o -o a.co
//   hipcc -O2 -o run_instrumented run_instrumented.hip && ./run_instrumented a.co
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef unsigned long long u64;
typedef unsigned int u32;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

int main(int argc, char **argv) {
    if (argc < 2) { printf("usage: %s code-object\n", argv[0]); return 2; }
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
    const u64 total = n * wpr;
    std::vector<u32> want(total, 0);
    for (u64 r = 0; r < n; r++)
        for (u64 i = 0; i < off[r + 1] - off[r]; i++) {
            u32 c = table[seq[off[r] + i]], st = (c & 0x80) ? 0 : c;
            u64 bit = i * 5;
            want[r * wpr + bit / 32] |= st << (bit % 32);
            if (bit % 32 > 27) want[r * wpr + bit / 32 + 1] |= st >> (32 - bit % 32);
        }
    unsigned char *d_seq, *d_tab; u64 *d_off; u32 *d_packed, *d_lens, *d_flags; unsigned char *d_dbg;
    const u64 dbg_bytes = 0x100000ull + total * 32;
    CK(hipMalloc(&d_seq, seq.size())); CK(hipMalloc(&d_tab, 256)); CK(hipMalloc(&d_off, (n + 1) * 8));
    CK(hipMalloc(&d_packed, total * 4)); CK(hipMalloc(&d_lens, n * 4)); CK(hipMalloc(&d_flags, n * 4)); CK(hipMalloc(&d_dbg, dbg_bytes));
    CK(hipMemcpy(d_seq, seq.data(), seq.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(d_tab, table.data(), 256, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_off, off.data(), (n + 1) * 8, hipMemcpyHostToDevice));
    std::vector<u32> got(total), dbg(total * 8);
    for (int ai = 1; ai < argc; ai++)
    for (int trial = 0; trial < 2; trial++) {
        hipModule_t mod; hipFunction_t fn;
        CK(hipModuleLoad(&mod, argv[ai]));
        CK(hipModuleGetFunction(&fn, mod, "_Z12pack_variantILi65EEvPKhPKyyjS1_jPjS4_S4_S4_"));
        const bool instrumented = strstr(argv[ai], "instr") != nullptr;
        CK(hipMemset(d_flags, 0, n * 4)); CK(hipMemset(d_packed, 0xEE, total * 4)); CK(hipMemset(d_dbg, 0xAB, dbg_bytes));
        u64 n_reads = n; u32 wpr_ = wpr, k_ = k;
        const unsigned char *a0 = d_seq; const u64 *a1 = d_off; const unsigned char *a4 = d_tab; u32 *a6 = d_packed, *a7 = d_lens, *a8 = d_flags, *a9 = (u32 *)d_dbg;
        void *args[] = {&a0, &a1, &n_reads, &wpr_, &a4, &k_, &a6, &a7, &a8, &a9};
        CK(hipModuleLaunchKernel(fn, (unsigned)((total + 255) / 256), 1, 1, 256, 1, 1, 0, 0, args, nullptr));
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(got.data(), d_packed, total * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(dbg.data(), d_dbg + 0x100000, total * 32, hipMemcpyDeviceToHost));
        if (!instrumented) {
            u64 bad = 0;
            for (u64 t = 0; t < total; t++) bad += got[t] != want[t];
            printf("%-40s trial %d: %llu wrong words\n", argv[ai], trial, bad);
            CK(hipModuleUnload(mod));
            continue;
        }
        u64 bad = 0, vcc_clear = 0, vcc_set_sel_zero = 0, vcc_set_sel_ok = 0, exec_missing = 0, shown = 0, shift_regs_wrong = 0;
        for (u64 t = 0; t < total; t++) {
            if (got[t] == want[t]) continue;
            bad++;
            const u32 *rec = &dbg[t * 8];  // latched in iteration 0: {vcc_lo, vcc_hi, lshl.lo, acc after the iteration, shift, -shift, iterations run, 0}
            const u64 vcc = rec[0] | ((u64)rec[1] << 32), exec = ~0ull;
            const u32 lane = (u32)(t & 63);
            const int shift = (int)rec[4];
            const u32 w = (u32)(t % wpr);
            const int want_shift = (int)(((32u * w) / 5) * 5) - (int)(32u * w);
            if (shift != want_shift || (int)rec[5] != -want_shift) shift_regs_wrong++;
            if (!((exec >> lane) & 1)) exec_missing++;
            const bool bit = (vcc >> lane) & 1;
            if (!bit) vcc_clear++;
            else if (rec[3] == 0) vcc_set_sel_zero++;
            else vcc_set_sel_ok++;
            if (shown < 6) {
                shown++;
                printf("  word %llu (wave %llu lane %u, word %u of its read): got %08x want %08x | iteration 0: shift %d (-shift reg %d), vcc %016llx (lane bit %d), "
                       "lshl.lo %08x, acc after iteration 0 %08x, iterations %u\n", t, t >> 6, lane, w, got[t], want[t], shift, (int)rec[5], vcc, (int)bit, rec[2], rec[3], rec[6]);
            }
        }
        printf("%s trial %d: %llu wrong words; of these: lane's vcc bit CLEAR after v_cmp_gt_i32(0 > shift) although shift < 0: %llu; vcc bit set but select gave 0: %llu; "
               "vcc set and select non-zero: %llu; lane missing from exec: %llu; shift registers not the expected values: %llu\n",
               argv[ai], trial, bad, vcc_clear, vcc_set_sel_zero, vcc_set_sel_ok, exec_missing, shift_regs_wrong);
        // in waves with wrong words: is vcc wrong for the whole wave?  compare the recorded vcc with the one the shifts imply
        u64 waves_bad = 0, waves_vcc_zero = 0;
        for (u64 wv = 0; wv < total / 64; wv++) {
            bool any = false;
            for (u32 l = 0; l < 64; l++) any |= got[wv * 64 + l] != want[wv * 64 + l];
            if (!any) continue;
            waves_bad++;
            // vcc is wave-wide: every lane that ran the loop latched the same value
            u64 vcc = 0; bool found = false;
            for (u32 l = 0; l < 64 && !found; l++) { const u32 *r2 = &dbg[(wv * 64 + l) * 8]; if (r2[0] != 0xdeadbeefu) { vcc = r2[0] | ((u64)r2[1] << 32); found = true; } }
            if (found && vcc == 0) waves_vcc_zero++;
        }
        printf("         %llu waves hold wrong words; in %llu of them the vcc recorded after the compare is 0 for the whole wave\n", waves_bad, waves_vcc_zero);
    }
    return 0;
}
