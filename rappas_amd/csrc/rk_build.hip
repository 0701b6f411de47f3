// rk_build.hip -- phylo-kmer database construction on the GPU (rk_build_db of include/rappas_place.h).
//
// Stage 1 (explore_kernel, hand-written): one LANE per (node, pos) explorer.  The reference's recursion
// (src/core/algos/WordExplorer_v3.java:98-199) keeps ONE running float that is incremented on the way down and
// decremented on the way up, so the score a word is registered with depends on everything explored before it by the same
// explorer: an explorer is inherently sequential and is replayed statement by statement (iterative form, frames in LDS);
// the parallelism is across the n_nodes x (L-k+2) explorers, handed out dynamically (explorers differ by orders of
// magnitude in size).  Registered (code << 16 | branch, score) tuples are appended to one global buffer, one atomic per
// wave and step.
// Stage 2 (library primitives, rocPRIM through hipCUB): radix sort by key, reduce-by-key with max
// (src/core/hash/CustomHash_v4_FastUtil81.java:73-89 keeps the largest PP* per (k-mer, branch)), run-length encode of the
// codes -> CSR.
// No CPU fallback: without a HIP device rk_build_db fails with RK_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <vector>

#include "../../include/rappas_place.h"
#include "rk_internal.h"

namespace rk {

typedef unsigned long long u64;
typedef unsigned int u32;

constexpr int BUILD_MAX_K = 16;        // frames per lane
constexpr int BUILD_WAVES_PER_BLOCK = 4;

struct BuildArgs {
    const unsigned char *states;  // [n_nodes][n_sites][n_states]
    const float *pp;
    const unsigned short *node_branch;
    const u32 *gap_off;
    const int *gap_len;
    u32 k, bits, n_nodes, n_sites, n_states, n_pos;  // n_pos = n_sites - k + 2 explorers per node
    u32 do_gap, limit1;
    float T;
    u64 n_tasks;
    u64 *task_counter;   // next explorer to hand out
    u64 *tuple_counter;  // tuples registered so far (keeps counting past `capacity`)
    u64 *visit_counter;
    u64 capacity;
    u64 *keys;           // [capacity]
    float *scores;       // [capacity]
};

// One frame per depth of the recursion (current_k): the site and pp of the node whose child loop is running, the loop
// variable j2 and the cursor of the gap sub-loop (-1: the plain child of this j2 has not been explored yet).
struct Frames {
    int site[BUILD_MAX_K][64];
    float p[BUILD_MAX_K][64];
    int j2[BUILD_MAX_K][64];
    int g[BUILD_MAX_K][64];
    int gend[BUILD_MAX_K][64];
};

__global__ void __launch_bounds__(64 * BUILD_WAVES_PER_BLOCK) explore_kernel(BuildArgs a) {
    __shared__ Frames frames[BUILD_WAVES_PER_BLOCK];
    const u32 lane = threadIdx.x & 63;
    Frames &F = frames[threadIdx.x >> 6];
    const int k = (int)a.k, S = (int)a.n_sites, NS = (int)a.n_states;

    // WordExplorer_v3 fields of the lane's current explorer
    float sum = 0.0f;
    bool bound = false;
    int boundK = -1, firstJump = -1;
    u64 code = 0;
    int d = -2;        // depth of the frame whose loop is running; -1 = the driver loop over first states; -2 = no explorer
    int top_j = 0;     // driver loop variable (Main_DBBUILD_3.java:710)
    int pos = 0;
    size_t node_base = 0;  // node * n_sites * n_states
    u32 branch = 0;
    u64 visits = 0;
    bool exhausted = false;

    while (true) {
        // fetch the next explorer (Main_DBBUILD_3.java:697-704: a fresh WordExplorer_v3); one atomic per wave
        const bool want = d == -2 && !exhausted;
        const u64 wm = __ballot(want);
        if (wm) {
            const int leader = __builtin_ctzll(wm);
            u64 base = 0;
            if ((int)lane == leader) base = atomicAdd(a.task_counter, (u64)__builtin_popcountll(wm));
            base = ((u64)(u32)__builtin_amdgcn_readlane((int)(u32)(base >> 32), leader) << 32) |
                   (u32)__builtin_amdgcn_readlane((int)(u32)base, leader);
            const u64 t = base + (u64)__builtin_popcountll(wm & ((1ull << lane) - 1));
            if (!want) {
            } else if (t >= a.n_tasks) {
                exhausted = true;
            } else {
                const u32 node = (u32)(t / a.n_pos);
                pos = (int)(t - (u64)node * a.n_pos);
                node_base = (size_t)node * a.n_sites * a.n_states;
                branch = a.node_branch[node];
                sum = 0.0f; bound = false; boundK = -1; firstJump = -1; code = 0;
                d = -1; top_j = 0;
            }
        }
        if (__all(exhausted)) break;

        // ---- one transition of the explorer: the next call (site ci, rank cj, depth cd), or leave a frame ----
        // Loop state of frame d: (j2, g).  g == -1: j2 not started (break test, then the plain child i+1);
        // g == -2: plain child done, gap jumps of this j2 still to be decided (gap mode only); g >= 0: gap cursor.
        // A call either returns at once (leaf, or a site beyond the alignment) or opens frame d+1; in both cases the
        // caller's loop state has already been advanced, so "return" needs no bookkeeping of its own.
        bool call = false;
        int ci = 0, cj = 0, cd = 0;
        if (d == -1) {
            if (top_j < NS) { call = true; ci = pos; cj = top_j; cd = 0; top_j++; }  // Main_DBBUILD_3.java:710-712
            else d = -2;  // explorer finished
        } else if (d >= 0) {
            const int i = F.site[d][lane];
            const int j2 = F.j2[d][lane], g = F.g[d][lane];
            if (g == -2) {  // WordExplorer_v3.java:161-186
                bool jump = false;
                if (i < S - 1) {
                    const int g0 = (int)a.gap_off[i + 1], g1 = (int)a.gap_off[i + 2];
                    if (g1 > g0) {
                        if (!a.limit1) jump = true;
                        else if (firstJump == -1) { firstJump = i; jump = true; }
                        if (jump) { F.g[d][lane] = g0; F.gend[d][lane] = g1; }
                    }
                }
                if (!jump) { F.g[d][lane] = -1; F.j2[d][lane] = j2 + 1; }
            } else if (g >= 0) {
                if (g < F.gend[d][lane]) {
                    call = true; ci = (i + 1) + a.gap_len[g]; cj = j2; cd = d + 1;
                    F.g[d][lane] = g + 1;
                } else {
                    F.g[d][lane] = -1;
                    F.j2[d][lane] = j2 + 1;
                }
            } else if (j2 >= NS || (bound && boundK == d + 1)) {  // loop end / break (:147-150)
                const float p = F.p[d][lane];
                sum = (float)((double)sum - (double)p);  // :198
                d = d - 1;                               // back in the caller's loop (or the driver loop)
            } else {
                call = true; ci = i + 1; cj = j2; cd = d + 1;  // :155-157
                if (a.do_gap) F.g[d][lane] = -2;
                else F.j2[d][lane] = j2 + 1;
            }
        }

        // ---- exploreWords(ci, cj) at depth cd (:98-143): leaves return at once, inner nodes open a frame ----
        bool emit = false;
        float emit_score = 0.0f;
        if (call && ci <= S - 1) {  // :109-111
            if (cd == 0) firstJump = -1;  // :113-115
            const size_t at = node_base + (size_t)ci * NS + (size_t)cj;
            const u32 st = a.states[at];
            const float p = a.pp[at];
            visits++;
            const u32 sh = a.bits * (u32)cd;
            code = (code & ~(((1ull << a.bits) - 1) << sh)) | ((u64)st << sh);  // word[current_k] = state (:117)
            sum = (float)((double)sum + (double)p);                              // :119 float += double
            bound = sum < a.T;                                                   // :120
            if (bound) boundK = cd;                                              // :121-123
            if (cd == k - 1) {
                if (!bound) { emit = true; emit_score = sum; }                   // :128-138 addTuple
                sum = (float)((double)sum - (double)p);                          // :141
            } else {
                F.site[cd][lane] = ci; F.p[cd][lane] = p; F.j2[cd][lane] = 0; F.g[cd][lane] = -1;
                d = cd;  // its child loop runs next
            }
        }
        // ---- register tuples: one atomic per wave ----
        const u64 em = __ballot(emit);
        if (em) {
            const int n = __builtin_popcountll(em);
            const int leader = __builtin_ctzll(em);
            u64 base = 0;
            if ((int)lane == leader) base = atomicAdd(a.tuple_counter, (u64)n);
            base = ((u64)(u32)__builtin_amdgcn_readlane((int)(u32)(base >> 32), leader) << 32) |
                   (u32)__builtin_amdgcn_readlane((int)(u32)base, leader);
            if (emit) {
                const u64 slot = base + (u64)__builtin_popcountll(em & ((1ull << lane) - 1));
                if (slot < a.capacity) {
                    a.keys[slot] = (code << 16) | branch;
                    a.scores[slot] = emit_score;
                }
            }
        }
    }
    // per-wave visit count
    for (int s = 32; s > 0; s >>= 1) visits += __shfl_xor(visits, s, 64);
    if (lane == 0 && visits) atomicAdd(a.visit_counter, visits);
}

struct ShiftRight16 {
    __host__ __device__ __forceinline__ u64 operator()(const u64 &k) const { return k >> 16; }
};

}  // namespace rk

using namespace rk;

namespace {
struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    int alloc(size_t bytes) {
        if (p) { (void)hipFree(p); p = nullptr; }
        hipError_t e = hipMalloc(&p, bytes ? bytes : 1);
        if (e != hipSuccess) { p = nullptr; return fail_msg(e == hipErrorOutOfMemory ? RK_ERR_NOMEM : RK_ERR_HIP, "rk_build_db: hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e)); }
        return RK_OK;
    }
    template <class T> T *as() { return (T *)p; }
};
}  // namespace

extern "C" void rk_built_free(rk_built_db *b) {
    if (!b) return;
    free(b->key_codes); free(b->row_offsets); free(b->branch_ids); free(b->scores);
    memset(b, 0, sizeof(*b));
}

extern "C" int rk_build_db(const rk_build_desc *d, rk_built_db *out) {
    if (!d || !out) return fail_msg(RK_ERR_INVALID, "rk_build_db: null argument");
    memset(out, 0, sizeof(*out));
    if (d->alphabet != RK_ALPHABET_DNA && d->alphabet != RK_ALPHABET_AA)
        return fail_msg(RK_ERR_INVALID, "rk_build_db: alphabet must be 4 (DNA) or 20 (AA), got %u", d->alphabet);
    const u32 bits = d->alphabet == RK_ALPHABET_DNA ? 2 : 5;
    const u32 kmax = d->alphabet == RK_ALPHABET_DNA ? 15 : 9;
    if (d->k < 2 || d->k > kmax) return fail_msg(RK_ERR_UNSUPPORTED, "rk_build_db: k=%u outside supported range 2..%u for this alphabet", d->k, kmax);
    if (d->n_states < 1 || d->n_states > (1u << bits) || d->n_states > d->alphabet)
        return fail_msg(RK_ERR_INVALID, "rk_build_db: n_states=%u does not fit the alphabet", d->n_states);
    if (d->n_sites < 1 || d->n_sites > (1u << 30)) return fail_msg(RK_ERR_INVALID, "rk_build_db: n_sites=%u out of range", d->n_sites);
    if (!std::isfinite(d->thr_log10)) return fail_msg(RK_ERR_INVALID, "rk_build_db: thr_log10 must be finite");
    if (d->n_nodes && (!d->states || !d->pp_log10 || !d->node_branch)) return fail_msg(RK_ERR_INVALID, "rk_build_db: null table");
    if (d->do_gap_jumps && (!d->gap_off || (d->gap_off[d->n_sites] && !d->gap_len)))
        return fail_msg(RK_ERR_INVALID, "rk_build_db: gap jumps requested without gap intervals");
    const size_t cells = (size_t)d->n_nodes * d->n_sites * d->n_states;
    for (size_t i = 0; i < cells; i++) {
        if (d->states[i] >= d->alphabet) return fail_msg(RK_ERR_INVALID, "rk_build_db: state %u at cell %zu is not a state of the alphabet", d->states[i], i);
        if (std::isnan(d->pp_log10[i])) return fail_msg(RK_ERR_INVALID, "rk_build_db: NaN posterior at cell %zu", i);
    }
    size_t n_gap = 0;
    if (d->do_gap_jumps) {
        for (u32 i = 0; i < d->n_sites; i++)
            if (d->gap_off[i + 1] < d->gap_off[i]) return fail_msg(RK_ERR_INVALID, "rk_build_db: gap_off not monotone at site %u", i);
        n_gap = d->gap_off[d->n_sites];
        for (size_t g = 0; g < n_gap; g++)
            if (d->gap_len[g] < 1) return fail_msg(RK_ERR_INVALID, "rk_build_db: gap interval %zu has length %d", g, d->gap_len[g]);
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail_msg(RK_ERR_NO_DEVICE, "rk_build_db: no HIP device available (no CPU fallback)");
    if (d->device < 0 || d->device >= ndev) return fail_msg(RK_ERR_INVALID, "rk_build_db: device %d out of range", d->device);
    RK_HIP_TRY(hipSetDevice(d->device));
    hipDeviceProp_t prop;
    RK_HIP_TRY(hipGetDeviceProperties(&prop, d->device));

    const u32 n_pos = d->n_sites + 2 > d->k ? d->n_sites - d->k + 2 : 0;  // Main_DBBUILD_3.java:693
    const u64 n_tasks = (u64)d->n_nodes * n_pos;
    DevBuf b_states, b_pp, b_nb, b_goff, b_glen, b_cnt;
    int rc;
    if ((rc = b_states.alloc(cells)) || (rc = b_pp.alloc(cells * 4)) || (rc = b_nb.alloc((size_t)d->n_nodes * 2)) ||
        (rc = b_goff.alloc(((size_t)d->n_sites + 2) * 4)) || (rc = b_glen.alloc(n_gap * 4)) || (rc = b_cnt.alloc(3 * 8)))
        return rc;
    RK_HIP_TRY(hipMemcpy(b_states.p, d->states, cells, hipMemcpyHostToDevice));
    RK_HIP_TRY(hipMemcpy(b_pp.p, d->pp_log10, cells * 4, hipMemcpyHostToDevice));
    RK_HIP_TRY(hipMemcpy(b_nb.p, d->node_branch, (size_t)d->n_nodes * 2, hipMemcpyHostToDevice));
    if (d->do_gap_jumps) {
        std::vector<u32> goff(d->gap_off, d->gap_off + d->n_sites + 1);
        goff.push_back(goff.back());  // gap_off[i + 2] is read for i + 1 == n_sites - 1
        RK_HIP_TRY(hipMemcpy(b_goff.p, goff.data(), goff.size() * 4, hipMemcpyHostToDevice));
        if (n_gap) RK_HIP_TRY(hipMemcpy(b_glen.p, d->gap_len, n_gap * 4, hipMemcpyHostToDevice));
    }

    BuildArgs a;
    memset(&a, 0, sizeof(a));
    a.states = b_states.as<unsigned char>(); a.pp = b_pp.as<float>(); a.node_branch = b_nb.as<unsigned short>();
    a.gap_off = b_goff.as<u32>(); a.gap_len = b_glen.as<int>();
    a.k = d->k; a.bits = bits; a.n_nodes = d->n_nodes; a.n_sites = d->n_sites; a.n_states = d->n_states; a.n_pos = n_pos;
    a.do_gap = d->do_gap_jumps ? 1u : 0u; a.limit1 = d->limit_to_1_jump ? 1u : 0u;
    a.T = d->thr_log10; a.n_tasks = n_tasks;
    a.task_counter = b_cnt.as<u64>(); a.tuple_counter = a.task_counter + 1; a.visit_counter = a.task_counter + 2;

    hipEvent_t e0, e1, e2;
    RK_HIP_TRY(hipEventCreate(&e0)); RK_HIP_TRY(hipEventCreate(&e1)); RK_HIP_TRY(hipEventCreate(&e2));
    struct EvGuard { hipEvent_t a, b, c; ~EvGuard() { (void)hipEventDestroy(a); (void)hipEventDestroy(b); (void)hipEventDestroy(c); } } evg{e0, e1, e2};

    // ---- stage 1: explore; the tuple buffer is sized by a guess first and by the exact count if the guess was short ----
    DevBuf b_keys, b_scores;
    u64 capacity = n_tasks * 64 + (1u << 20);
    if (capacity > (1ull << 28)) capacity = 1ull << 28;
    u64 counters[3] = {0, 0, 0};
    const unsigned blocks = (unsigned)prop.multiProcessorCount * 4;  // persistent: 16 waves per CU (LDS: 20 KB per wave)
    float explore_ms = 0.0f;
    for (int attempt = 0; attempt < 2; attempt++) {
        if ((rc = b_keys.alloc(capacity * 8)) || (rc = b_scores.alloc(capacity * 4))) return rc;
        a.capacity = capacity; a.keys = b_keys.as<u64>(); a.scores = b_scores.as<float>();
        RK_HIP_TRY(hipMemset(b_cnt.p, 0, 3 * 8));
        RK_HIP_TRY(hipEventRecord(e0, 0));
        if (n_tasks) hipLaunchKernelGGL(explore_kernel, dim3(blocks), dim3(64 * BUILD_WAVES_PER_BLOCK), 0, 0, a);
        RK_HIP_TRY(hipGetLastError());
        RK_HIP_TRY(hipEventRecord(e1, 0));
        RK_HIP_TRY(hipEventSynchronize(e1));
        RK_HIP_TRY(hipEventElapsedTime(&explore_ms, e0, e1));
        RK_HIP_TRY(hipMemcpy(counters, b_cnt.p, sizeof(counters), hipMemcpyDeviceToHost));
        if (counters[1] <= capacity) break;
        if (attempt == 1) return fail_msg(RK_ERR_HIP, "rk_build_db: tuple count changed between two identical launches");
        capacity = counters[1];
    }
    const u64 n_tuples = counters[1];
    out->tuples = n_tuples;
    out->visits = counters[2];
    out->explore_ms = explore_ms;
    if (n_tuples >= (1ull << 31)) return fail_msg(RK_ERR_UNSUPPORTED, "rk_build_db: %llu tuples in one call; split the nodes into batches", (unsigned long long)n_tuples);

    // ---- stage 2: sort by (code, branch), max per key, run lengths per code ----
    RK_HIP_TRY(hipEventRecord(e1, 0));
    u64 n_entries = 0, n_keys = 0;
    std::vector<u64> h_keys_unique;
    std::vector<float> h_scores;
    std::vector<u64> h_codes;
    std::vector<int> h_counts;
    if (n_tuples) {
        const int n = (int)n_tuples;
        DevBuf b_keys2, b_scores2, b_tmp, b_nruns, b_codes, b_counts;
        if ((rc = b_keys2.alloc((size_t)n * 8)) || (rc = b_scores2.alloc((size_t)n * 4)) || (rc = b_nruns.alloc(8))) return rc;
        hipcub::DoubleBuffer<u64> kb(b_keys.as<u64>(), b_keys2.as<u64>());
        hipcub::DoubleBuffer<float> vb(b_scores.as<float>(), b_scores2.as<float>());
        const int end_bit = (int)(16 + bits * d->k);
        size_t tmp_bytes = 0;
        RK_HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, kb, vb, n, 0, end_bit));
        if ((rc = b_tmp.alloc(tmp_bytes))) return rc;
        RK_HIP_TRY(hipcub::DeviceRadixSort::SortPairs(b_tmp.p, tmp_bytes, kb, vb, n, 0, end_bit));
        u64 *sorted_k = kb.Current();
        float *sorted_v = vb.Current();
        u64 *uniq_k = kb.Alternate();
        float *max_v = vb.Alternate();
        tmp_bytes = 0;
        RK_HIP_TRY(hipcub::DeviceReduce::ReduceByKey(nullptr, tmp_bytes, sorted_k, uniq_k, sorted_v, max_v, b_nruns.as<int>(), hipcub::Max(), n));
        if ((rc = b_tmp.alloc(tmp_bytes))) return rc;
        RK_HIP_TRY(hipcub::DeviceReduce::ReduceByKey(b_tmp.p, tmp_bytes, sorted_k, uniq_k, sorted_v, max_v, b_nruns.as<int>(), hipcub::Max(), n));
        int runs = 0;
        RK_HIP_TRY(hipMemcpy(&runs, b_nruns.p, 4, hipMemcpyDeviceToHost));
        n_entries = (u64)runs;
        // run lengths of the codes over the unique (code, branch) keys
        if ((rc = b_codes.alloc((size_t)runs * 8)) || (rc = b_counts.alloc((size_t)runs * 4))) return rc;
        hipcub::TransformInputIterator<u64, ShiftRight16, u64 *> code_it(uniq_k, ShiftRight16());
        tmp_bytes = 0;
        RK_HIP_TRY(hipcub::DeviceRunLengthEncode::Encode(nullptr, tmp_bytes, code_it, b_codes.as<u64>(), b_counts.as<int>(), b_nruns.as<int>(), runs));
        if ((rc = b_tmp.alloc(tmp_bytes))) return rc;
        RK_HIP_TRY(hipcub::DeviceRunLengthEncode::Encode(b_tmp.p, tmp_bytes, code_it, b_codes.as<u64>(), b_counts.as<int>(), b_nruns.as<int>(), runs));
        int nk = 0;
        RK_HIP_TRY(hipMemcpy(&nk, b_nruns.p, 4, hipMemcpyDeviceToHost));
        n_keys = (u64)nk;
        RK_HIP_TRY(hipEventRecord(e2, 0));
        RK_HIP_TRY(hipEventSynchronize(e2));
        float ms = 0.0f;
        RK_HIP_TRY(hipEventElapsedTime(&ms, e1, e2));
        out->reduce_ms = ms;
        h_keys_unique.resize(n_entries); h_scores.resize(n_entries); h_codes.resize(n_keys); h_counts.resize(n_keys);
        RK_HIP_TRY(hipMemcpy(h_keys_unique.data(), uniq_k, n_entries * 8, hipMemcpyDeviceToHost));
        RK_HIP_TRY(hipMemcpy(h_scores.data(), max_v, n_entries * 4, hipMemcpyDeviceToHost));
        RK_HIP_TRY(hipMemcpy(h_codes.data(), b_codes.p, n_keys * 8, hipMemcpyDeviceToHost));
        RK_HIP_TRY(hipMemcpy(h_counts.data(), b_counts.p, n_keys * 4, hipMemcpyDeviceToHost));
    }
    out->n_keys = n_keys;
    out->n_entries = n_entries;
    out->key_codes = (uint64_t *)malloc((n_keys + 1) * 8);
    out->row_offsets = (uint64_t *)malloc((n_keys + 1) * 8);
    out->branch_ids = (uint16_t *)malloc((n_entries + 1) * 2);
    out->scores = (float *)malloc((n_entries + 1) * 4);
    if (!out->key_codes || !out->row_offsets || !out->branch_ids || !out->scores) {
        rk_built_free(out);
        return fail_msg(RK_ERR_NOMEM, "rk_build_db: host OOM for %llu entries", (unsigned long long)n_entries);
    }
    u64 acc = 0;
    for (u64 i = 0; i < n_keys; i++) {
        out->key_codes[i] = h_codes[i];
        out->row_offsets[i] = acc;
        acc += (u64)h_counts[i];
    }
    out->row_offsets[n_keys] = acc;
    for (u64 i = 0; i < n_entries; i++) {
        out->branch_ids[i] = (uint16_t)(h_keys_unique[i] & 0xFFFF);
        out->scores[i] = h_scores[i];
    }
    return RK_OK;
}
