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
#include <cstring>
#include <rocprim/rocprim.hpp>

#include <algorithm>
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
    u64 n_tasks, task_base;  // explorers of this launch: task_base .. task_base + n_tasks (node batches)
    u64 *task_counter;   // next explorer to hand out
    u64 *slot_counter;   // tuple-buffer slots handed out so far, in whole chunks (keeps counting past `capacity`)
    u64 *visit_counter;
    u64 *tuple_counter;  // addTuple calls
    u64 capacity;
    u64 *keys;           // [capacity]
    float *scores;       // [capacity]
};

// One frame per depth of the recursion (current_k): the site and pp of the node whose child loop is running, and the loop
// state packed in one word: j2 (bits 0-4) and the gap cursor g + 2 (bits 5-31; g == -1: the plain child of this j2 has not
// been explored yet, g == -2: it has, gap jumps undecided).  12 bytes per frame and lane: 10.5 KB of LDS per wave.
constexpr int BUILD_FRAMES = 14;  // inner nodes live at depths 0 .. k-2 <= 13
struct Frames {
    int site[BUILD_FRAMES][64];
    float p[BUILD_FRAMES][64];
    int j2g[BUILD_FRAMES][64];
};
__device__ __forceinline__ int pack_j2g(int j2, int g) { return j2 | ((g + 2) << 5); }

// addTuple (CustomHash_v4_FastUtil81.java:73): append (code << 16 | branch, score).  A wave owns a CHUNK of the tuple
// buffer at a time (one atomic per chunk, none per registration: with a 20-letter alphabet one step in two registers
// something somewhere in the wave); the unused tail of a chunk is filled with TUPLE_PAD, which sorts behind every real key
// (no real key ends in 0xFFFF: branch ids are < 65535).  Any subset of the wave may be here, so the wave's cursor lives in
// LDS, not in registers.
constexpr int TUPLE_CHUNK = 1024;
constexpr u64 TUPLE_PAD = ~0ull;
struct WaveCursor {
    u64 base;
    int used;
    int pad_;
};

__device__ __forceinline__ void append_tuples(const BuildArgs &a, volatile WaveCursor *wc, bool emit, u64 key, float score,
                                              u32 lane, u64 &n_tuples) {
    const u64 em = __ballot(emit);
    if (!em) return;
    const int n = __builtin_popcountll(em);
    const int leader = __builtin_ctzll(em);
    const int rank = __builtin_popcountll(em & ((1ull << lane) - 1));
    u64 base = wc->base;
    int used = wc->used;
    if (used + n > TUPLE_CHUNK) {  // uniform: close this chunk (fewer than n <= 64 slots are left), open the next one
        if (emit && used + rank < TUPLE_CHUNK && base + used + rank < a.capacity) a.keys[base + used + rank] = TUPLE_PAD;
        u64 nb = 0;
        if ((int)lane == leader) nb = atomicAdd(a.slot_counter, (u64)TUPLE_CHUNK);
        base = ((u64)(u32)__builtin_amdgcn_readlane((int)(u32)(nb >> 32), leader) << 32) |
               (u32)__builtin_amdgcn_readlane((int)(u32)nb, leader);
        used = 0;
        if ((int)lane == leader) wc->base = base;
    }
    if (emit) {
        const u64 slot = base + (u64)(used + rank);
        if (slot < a.capacity) {
            a.keys[slot] = key;
            a.scores[slot] = score;
        }
        n_tuples++;
    }
    if ((int)lane == leader) wc->used = used + n;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

struct Explorer {  // WordExplorer_v3 fields (:37-55) of the lane's current explorer
    float sum;
    bool bound;
    int boundK;
    u64 code;
    u64 visits;
    u64 tuples;
    int firstJump;  // idxOfFirstJump (:49)
};

// exploreWords(site, rank) at depth `depth` for a node with exactly L levels below it (L == 0: a leaf), without gap jumps:
// WordExplorer_v3.java:98-157,198 as straight-line nested loops in registers -- the bottom of every explorer, where nearly
// all visits happen.  Same statements in the same order as the frame machine of explore_kernel.
template <int L>
__device__ __forceinline__ void explore_tail(const BuildArgs &a, volatile WaveCursor *wc, Explorer &e, int site, int rank,
                                             int depth, size_t node_base, u32 branch, u32 lane) {
    if (site > (int)a.n_sites - 1) return;                                  // :109-111
    const size_t at = node_base + (size_t)site * a.n_states + (size_t)rank;
    const u32 st = a.states[at];
    const float p = a.pp[at];
    e.visits++;
    const u32 sh = a.bits * (u32)depth;
    e.code = (e.code & ~(((1ull << a.bits) - 1) << sh)) | ((u64)st << sh);  // :117
    e.sum = (float)((double)e.sum + (double)p);                             // :119
    e.bound = e.sum < a.T;                                                  // :120
    if (e.bound) e.boundK = depth;                                          // :121-123
    if (L == 0) {
        append_tuples(a, wc, !e.bound, (e.code << 16) | branch, e.sum, lane, e.tuples);  // :128-138
        e.sum = (float)((double)e.sum - (double)p);                         // :141
        return;
    } else {
        for (int j2 = 0; j2 < (int)a.n_states; j2++) {                      // :147
            if (e.bound && e.boundK == depth + 1) break;                    // :148-150
            explore_tail<(L > 0 ? L - 1 : 0)>(a, wc, e, site + 1, j2, depth + 1, node_base, branch, lane);
        }
        e.sum = (float)((double)e.sum - (double)p);                         // :198
    }
}

// The same with gap jumps (:161-186): after the plain child at site + 1, the children at site + 1 + len for every gap interval
// that starts at site + 1 -- all of them, or (limitTo1Jump) only while no jump has been taken since the explorer's current first
// state (`firstJump` is sticky, exactly like idxOfFirstJump).  Jumped-to sites are arbitrary, so every visit loads its cell.
template <int L>
__device__ __forceinline__ void explore_tail_gaps(const BuildArgs &a, volatile WaveCursor *wc, Explorer &e, int site, int rank,
                                                  int depth, size_t node_base, u32 branch, u32 lane) {
    if (site > (int)a.n_sites - 1) return;                                  // :109-111
    if (depth == 0) e.firstJump = -1;                                       // :113-115
    const size_t at = node_base + (size_t)site * a.n_states + (size_t)rank;
    const u32 st = a.states[at];
    const float p = a.pp[at];
    e.visits++;
    const u32 sh = a.bits * (u32)depth;
    e.code = (e.code & ~(((1ull << a.bits) - 1) << sh)) | ((u64)st << sh);  // :117
    e.sum = (float)((double)e.sum + (double)p);                             // :119
    e.bound = e.sum < a.T;                                                  // :120
    if (e.bound) e.boundK = depth;                                          // :121-123
    if (L == 0) {
        append_tuples(a, wc, !e.bound, (e.code << 16) | branch, e.sum, lane, e.tuples);  // :128-138
        e.sum = (float)((double)e.sum - (double)p);                         // :141
        return;
    } else {
        for (int j2 = 0; j2 < (int)a.n_states; j2++) {                      // :147
            if (e.bound && e.boundK == depth + 1) break;                    // :148-150
            explore_tail_gaps<(L > 0 ? L - 1 : 0)>(a, wc, e, site + 1, j2, depth + 1, node_base, branch, lane);
            if (site < (int)a.n_sites - 1) {                                // :161
                const int g0 = (int)a.gap_off[site + 1], g1 = (int)a.gap_off[site + 2];
                if (g1 > g0) {                                              // :163
                    bool jump = !a.limit1;
                    if (a.limit1 && e.firstJump == -1) { e.firstJump = site; jump = true; }  // :174-175
                    if (jump)
                        for (int g = g0; g < g1; g++)
                            explore_tail_gaps<(L > 0 ? L - 1 : 0)>(a, wc, e, (site + 1) + a.gap_len[g], j2, depth + 1, node_base, branch, lane);
                }
            }
        }
        e.sum = (float)((double)e.sum - (double)p);                         // :198
    }
}

// Four-state alphabets (DNA): the rows of the INL + 1 sites a tail touches are loaded once at its entry (one float4 of
// posteriors and one packed word of states per site) and every visit below reads registers -- no load, no load latency, per
// visit.  Same statements as explore_tail; `TOP - L` is the level of the node inside the tail, a compile-time register index.
struct Sites4 {
    float4 p[4];
    u32 st[4];
};
__device__ __forceinline__ float pick4(const float4 &v, int r) { return r == 0 ? v.x : (r == 1 ? v.y : (r == 2 ? v.z : v.w)); }

template <int TOP, int L>
__device__ __forceinline__ void explore_tail4(const BuildArgs &a, volatile WaveCursor *wc, Explorer &e, const Sites4 &t,
                                              int site, int rank, int depth, u32 branch, u32 lane) {
    if (site > (int)a.n_sites - 1) return;                                  // :109-111
    const u32 st = (t.st[TOP - L] >> (8 * rank)) & 0xFFu;
    const float p = pick4(t.p[TOP - L], rank);
    e.visits++;
    const u32 sh = 2u * (u32)depth;
    e.code = (e.code & ~(3ull << sh)) | ((u64)st << sh);                    // :117
    e.sum = (float)((double)e.sum + (double)p);                             // :119
    e.bound = e.sum < a.T;                                                  // :120
    if (e.bound) e.boundK = depth;                                          // :121-123
    if (L == 0) {
        append_tuples(a, wc, !e.bound, (e.code << 16) | branch, e.sum, lane, e.tuples);  // :128-138
        e.sum = (float)((double)e.sum - (double)p);                         // :141
        return;
    } else {
        for (int j2 = 0; j2 < 4; j2++) {                                    // :147
            if (e.bound && e.boundK == depth + 1) break;                    // :148-150
            explore_tail4<TOP, (L > 0 ? L - 1 : 0)>(a, wc, e, t, site + 1, j2, depth + 1, branch, lane);
        }
        e.sum = (float)((double)e.sum - (double)p);                         // :198
    }
}

template <int TOP>
__device__ __forceinline__ void run_tail4(const BuildArgs &a, volatile WaveCursor *wc, Explorer &e, int site, int rank, int depth,
                                          size_t node_base, u32 branch, u32 lane) {
    Sites4 t;
#pragma unroll
    for (int l = 0; l <= TOP; l++) {
        const int sl = min(site + l, (int)a.n_sites - 1);  // rows beyond the alignment are never read back (:109-111)
        const size_t at = node_base + (size_t)sl * 4;
        t.p[l] = *(const float4 *)(a.pp + at);
        t.st[l] = *(const u32 *)(a.states + at);
    }
    explore_tail4<TOP, TOP>(a, wc, e, t, site, rank, depth, branch, lane);
}

// Twenty-state alphabets (amino acids): 95 % of the visits of a tail are its leaves, which all sit on ONE site; that site's row
// (20 posteriors + 20 packed states) is loaded once at the tail's entry and the leaf loop, fully unrolled, reads registers.
// The 1 + 20 + 400 nodes above the leaves load per visit as in explore_tail.
struct Row20 {
    float p[20];
    u32 st[5];
};

template <int L>
__device__ __forceinline__ void explore_tail20(const BuildArgs &a, volatile WaveCursor *wc, Explorer &e, const Row20 &leaf,
                                               int site, int rank, int depth, size_t node_base, u32 branch, u32 lane) {
    if (site > (int)a.n_sites - 1) return;                                  // :109-111
    u32 st;
    float p;
    if (L == 0) {
        st = (leaf.st[rank >> 2] >> (8 * (rank & 3))) & 0xFFu;              // rank is a constant here (unrolled caller)
        p = leaf.p[rank];
    } else {
        const size_t at = node_base + (size_t)site * 20 + (size_t)rank;
        st = a.states[at];
        p = a.pp[at];
    }
    e.visits++;
    const u32 sh = 5u * (u32)depth;
    e.code = (e.code & ~(31ull << sh)) | ((u64)st << sh);                   // :117
    e.sum = (float)((double)e.sum + (double)p);                             // :119
    e.bound = e.sum < a.T;                                                  // :120
    if (e.bound) e.boundK = depth;                                          // :121-123
    if (L == 0) {
        append_tuples(a, wc, !e.bound, (e.code << 16) | branch, e.sum, lane, e.tuples);  // :128-138
        e.sum = (float)((double)e.sum - (double)p);                         // :141
        return;
    } else if (L == 1) {
#pragma unroll
        for (int j2 = 0; j2 < 20; j2++) {                                   // :147, unrolled: leaf.p[j2] is a register
            if (e.bound && e.boundK == depth + 1) break;                    // :148-150
            explore_tail20<0>(a, wc, e, leaf, site + 1, j2, depth + 1, node_base, branch, lane);
        }
        e.sum = (float)((double)e.sum - (double)p);                         // :198
    } else {
        for (int j2 = 0; j2 < 20; j2++) {
            if (e.bound && e.boundK == depth + 1) break;
            explore_tail20<(L > 1 ? L - 1 : 1)>(a, wc, e, leaf, site + 1, j2, depth + 1, node_base, branch, lane);
        }
        e.sum = (float)((double)e.sum - (double)p);
    }
}

template <int TOP>
__device__ __forceinline__ void run_tail20(const BuildArgs &a, volatile WaveCursor *wc, Explorer &e, int site, int rank, int depth,
                                           size_t node_base, u32 branch, u32 lane) {
    Row20 leaf;
    const int sl = min(site + TOP, (int)a.n_sites - 1);  // a leaf site beyond the alignment is never read back (:109-111)
    const size_t at = node_base + (size_t)sl * 20;
    const float4 *pv = (const float4 *)(a.pp + at);
    const u32 *sv = (const u32 *)(a.states + at);
#pragma unroll
    for (int q = 0; q < 5; q++) {
        const float4 v = pv[q];
        leaf.p[4 * q] = v.x; leaf.p[4 * q + 1] = v.y; leaf.p[4 * q + 2] = v.z; leaf.p[4 * q + 3] = v.w;
        leaf.st[q] = sv[q];
    }
    if (TOP == 0) {  // the called node is itself the leaf: its rank is not a compile-time constant
        if (site > (int)a.n_sites - 1) return;
        explore_tail<0>(a, wc, e, site, rank, depth, node_base, branch, lane);
    } else {
        explore_tail20<TOP>(a, wc, e, leaf, site, rank, depth, node_base, branch, lane);
    }
}

// INL = levels of the recursion below a node that explore_tail runs in registers (the node itself included: 1 + NS + ... +
// NS^INL visits per transition).
template <int INL, int VEC, bool GAPS>  // VEC: 4 / 20 = register-resident site rows for that many states, 0 = generic
__global__ void __launch_bounds__(64 * BUILD_WAVES_PER_BLOCK) explore_kernel(BuildArgs a) {
    __shared__ Frames frames[BUILD_WAVES_PER_BLOCK];
    __shared__ WaveCursor cursors[BUILD_WAVES_PER_BLOCK];
    const u32 lane = threadIdx.x & 63;
    Frames &F = frames[threadIdx.x >> 6];
    volatile WaveCursor *wc = &cursors[threadIdx.x >> 6];
    if (lane == 0) { wc->base = 0; wc->used = TUPLE_CHUNK; }  // the first registration opens the first chunk
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    constexpr bool V4 = VEC == 4, V20 = VEC == 20;
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
    u64 visits = 0, tuples = 0;
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
                const u64 tg = a.task_base + t;
                const u32 node = (u32)(tg / a.n_pos);
                pos = (int)(tg - (u64)node * a.n_pos);
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
            const int w = F.j2g[d][lane];
            const int j2 = w & 31, g = (w >> 5) - 2;
            if (g == -2) {  // WordExplorer_v3.java:161-186
                bool jump = false;
                if (i < S - 1) {
                    const int g0 = (int)a.gap_off[i + 1], g1 = (int)a.gap_off[i + 2];
                    if (g1 > g0) {
                        if (!a.limit1) jump = true;
                        else if (firstJump == -1) { firstJump = i; jump = true; }
                        if (jump) F.j2g[d][lane] = pack_j2g(j2, g0);
                    }
                }
                if (!jump) F.j2g[d][lane] = pack_j2g(j2 + 1, -1);
            } else if (g >= 0) {
                if (g < (int)a.gap_off[i + 2]) {  // end of gapIntervals[i + 1]
                    call = true; ci = (i + 1) + a.gap_len[g]; cj = j2; cd = d + 1;
                    F.j2g[d][lane] = pack_j2g(j2, g + 1);
                } else {
                    F.j2g[d][lane] = pack_j2g(j2 + 1, -1);
                }
            } else if (j2 >= NS || (bound && boundK == d + 1)) {  // loop end / break (:147-150)
                const float p = F.p[d][lane];
                sum = (float)((double)sum - (double)p);  // :198
                d = d - 1;                               // back in the caller's loop (or the driver loop)
            } else {
                call = true; ci = i + 1; cj = j2; cd = d + 1;  // :155-157
                F.j2g[d][lane] = a.do_gap ? pack_j2g(j2, -2) : pack_j2g(j2 + 1, -1);
            }
        }

        // ---- exploreWords(ci, cj) at depth cd (:98-143) ----
        // A node with at most INL levels below it is explored to the end right here (explore_tail);
        // higher nodes open a frame and their child loop runs through the transitions above.
        bool emit = false;
        float emit_score = 0.0f;
        const int below = k - 1 - cd;  // levels below the called node
        if (cd == 0 && call) firstJump = -1;  // :113-115 (a call at depth 0 beyond the alignment cannot happen: pos < n_sites)
        // one-jump mode: once the explorer has jumped for its current first state, the rest of that first state's subtree is
        // explored exactly as without gap jumps (idxOfFirstJump stays set): the register-resident tails apply again
        // ... and a subtree none of whose inner sites is followed by the start of a gap interval cannot jump in either mode
        bool jumps_possible = GAPS && call && below <= INL && !(a.limit1 && firstJump != -1);
        if (jumps_possible) {
            const int s_lo = min(ci + 1, S), s_hi = min(ci + below, S - 1) + 1;  // intervals starting in [ci + 1, ci + below]
            jumps_possible = s_hi > s_lo && a.gap_off[s_hi] > a.gap_off[s_lo];
        }
        constexpr int INL_GAPS = INL < 2 ? INL : 2;
        if (call && jumps_possible && below <= INL_GAPS) {
            Explorer e{sum, bound, boundK, code, visits, tuples, firstJump};
            switch (below) {
            case 0: explore_tail_gaps<0>(a, wc, e, ci, cj, cd, node_base, branch, lane); break;
            case 1: explore_tail_gaps<1>(a, wc, e, ci, cj, cd, node_base, branch, lane); break;
            case 2: if (INL_GAPS >= 2) explore_tail_gaps<(INL_GAPS >= 2 ? 2 : 0)>(a, wc, e, ci, cj, cd, node_base, branch, lane); break;
            default: break;
            }
            sum = e.sum; bound = e.bound; boundK = e.boundK; code = e.code; visits = e.visits; tuples = e.tuples;
            firstJump = e.firstJump;
        } else if (call && !jumps_possible && below <= INL) {
            Explorer e{sum, bound, boundK, code, visits, tuples, firstJump};
            if (V4) {
                switch (below) {
                case 0: run_tail4<0>(a, wc, e, ci, cj, cd, node_base, branch, lane); break;
                case 1: run_tail4<1>(a, wc, e, ci, cj, cd, node_base, branch, lane); break;
                case 2: if (INL >= 2) run_tail4<(INL >= 2 ? 2 : 0)>(a, wc, e, ci, cj, cd, node_base, branch, lane); break;
                case 3: if (INL >= 3) run_tail4<(INL >= 3 ? 3 : 0)>(a, wc, e, ci, cj, cd, node_base, branch, lane); break;
                default: break;
                }
            } else if (V20) {
                switch (below) {
                case 0: explore_tail<0>(a, wc, e, ci, cj, cd, node_base, branch, lane); break;
                case 1: run_tail20<1>(a, wc, e, ci, cj, cd, node_base, branch, lane); break;
                case 2: if (INL >= 2) run_tail20<(INL >= 2 ? 2 : 1)>(a, wc, e, ci, cj, cd, node_base, branch, lane); break;
                case 3: if (INL >= 3) run_tail20<(INL >= 3 ? 3 : 1)>(a, wc, e, ci, cj, cd, node_base, branch, lane); break;
                default: break;
                }
            } else {
                switch (below) {
                case 0: explore_tail<0>(a, wc, e, ci, cj, cd, node_base, branch, lane); break;
                case 1: explore_tail<1>(a, wc, e, ci, cj, cd, node_base, branch, lane); break;
                case 2: if (INL >= 2) explore_tail<(INL >= 2 ? 2 : 0)>(a, wc, e, ci, cj, cd, node_base, branch, lane); break;
                case 3: if (INL >= 3) explore_tail<(INL >= 3 ? 3 : 0)>(a, wc, e, ci, cj, cd, node_base, branch, lane); break;
                default: break;
                }
            }
            sum = e.sum; bound = e.bound; boundK = e.boundK; code = e.code; visits = e.visits; tuples = e.tuples;
        } else if (call && ci <= S - 1) {  // :109-111
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
                F.site[cd][lane] = ci; F.p[cd][lane] = p; F.j2g[cd][lane] = pack_j2g(0, -1);
                d = cd;  // its child loop runs next
            }
        }
        append_tuples(a, wc, emit, (code << 16) | branch, emit_score, lane, tuples);
    }
    // pad the unused tail of the wave's last chunk
    {
        const u64 base = wc->base;
        for (int i = wc->used + (int)lane; i < TUPLE_CHUNK; i += 64)
            if (base + (u64)i < a.capacity) a.keys[base + (u64)i] = TUPLE_PAD;
    }
    // per-wave visit count
    for (int s = 32; s > 0; s >>= 1) visits += __shfl_xor(visits, s, 64);
    for (int s = 32; s > 0; s >>= 1) tuples += __shfl_xor(tuples, s, 64);
    if (lane == 0 && visits) atomicAdd(a.visit_counter, visits);
    if (lane == 0 && tuples) atomicAdd(a.tuple_counter, tuples);
}

__global__ void split_keys_kernel(const u64 *keys, u64 n, unsigned short *branch) {
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x)
        branch[i] = (unsigned short)(keys[i] & 0xFFFFu);
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

static int build_db_impl(const rk_build_desc *d, rk_built_db *out);

extern "C" int rk_build_db(const rk_build_desc *d, rk_built_db *out) {
    RK_GUARD_BEGIN
    if (!d || !out) return fail_msg(RK_ERR_INVALID, "rk_build_db: null argument");
    memset(out, 0, sizeof(*out));
    int prev = -1;
    (void)hipGetDevice(&prev);  // the caller's current device is put back whatever happens below
    const int rc = build_db_impl(d, out);
    if (rc != RK_OK) rk_built_free(out);  // nothing half-built is handed back
    if (prev >= 0) (void)hipSetDevice(prev);
    return rc;
    RK_GUARD_END("rk_build_db")
}

static int build_db_impl(const rk_build_desc *d, rk_built_db *out) {
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
    for (u32 i = 0; i < d->n_nodes; i++)
        if (d->node_branch[i] == 0xFFFFu) return fail_msg(RK_ERR_INVALID, "rk_build_db: node %u: branch id 65535 is reserved", i);
    size_t n_gap = 0;
    if (d->do_gap_jumps) {
        for (u32 i = 0; i < d->n_sites; i++)
            if (d->gap_off[i + 1] < d->gap_off[i]) return fail_msg(RK_ERR_INVALID, "rk_build_db: gap_off not monotone at site %u", i);
        n_gap = d->gap_off[d->n_sites];
        for (size_t g = 0; g < n_gap; g++)
            // (an interval longer than the alignment cannot come out of Alignment.getGapIntervals; the kernel adds it to a site index)
            if (d->gap_len[g] < 1 || (u32)d->gap_len[g] > d->n_sites)
                return fail_msg(RK_ERR_INVALID, "rk_build_db: gap interval %zu has length %d (alignment has %u sites)", g, d->gap_len[g], d->n_sites);
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
        (rc = b_goff.alloc(((size_t)d->n_sites + 2) * 4)) || (rc = b_glen.alloc(n_gap * 4)) || (rc = b_cnt.alloc(4 * 8)))
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
    a.T = d->thr_log10; a.n_tasks = n_tasks; a.task_base = 0;
    a.task_counter = b_cnt.as<u64>(); a.slot_counter = a.task_counter + 1; a.visit_counter = a.task_counter + 2;
    a.tuple_counter = a.task_counter + 3;

    hipEvent_t e0, e1, e2;
    RK_HIP_TRY(hipEventCreate(&e0)); RK_HIP_TRY(hipEventCreate(&e1)); RK_HIP_TRY(hipEventCreate(&e2));
    struct EvGuard { hipEvent_t a, b, c; ~EvGuard() { (void)hipEventDestroy(a); (void)hipEventDestroy(b); (void)hipEventDestroy(c); } } evg{e0, e1, e2};

    // ---- node batches: explore -> sort -> max per key, folded into the running set of unique (key, best score) pairs ----
    // (one batch unless the tuple buffer of a single launch would pass ~2.7e8 slots: the sort is handed 32-bit item counts)
    const unsigned blocks = (unsigned)prop.multiProcessorCount * 3;  // persistent: 12 waves per CU (LDS: 10.5 KB per wave)
    const u64 chunk_slack = (u64)blocks * BUILD_WAVES_PER_BLOCK * TUPLE_CHUNK;  // every wave may leave one chunk half empty
    // slots one batch may fill: bounded by the 32-bit item counts of the sort and by device memory (12 bytes per slot, twice for
    // the sort's double buffers, plus its scratch) -- with 288 GB of HBM that is normally the whole input in one batch
    size_t mem_free = 0, mem_total = 0;
    RK_HIP_TRY(hipMemGetInfo(&mem_free, &mem_total));
    u64 target_slots = std::min<u64>((1ull << 31) - 2 - chunk_slack, (u64)mem_free / 40);
    if (target_slots < (1u << 22)) return fail_msg(RK_ERR_NOMEM, "rk_build_db: only %zu bytes of device memory free", mem_free);
    const int end_bit = (int)(16 + bits * d->k);
    DevBuf run_keys, run_vals;  // running result
    u64 run_n = 0;
    u64 total_tuples = 0, total_visits = 0;
    double explore_total = 0.0, reduce_total = 0.0;
    u32 batch = d->n_nodes;
    if (const char *e = rk_knob("RK_BUILD_BATCH_NODES")) batch = (u32)atoi(e);  // developer / test knob
    if (batch < 1) batch = 1;
    u32 node0 = 0;
    double slots_per_task = -1.0;  // measured on the batches done so far (explorers register 0 .. thousands of words each)
    const u64 generous = 4096;     // slots per explorer assumed while nothing has been measured
    while (node0 < d->n_nodes && n_pos) {
        u32 nb = std::min(batch, d->n_nodes - node0);
        // nothing measured yet and too many explorers to size the buffer generously: measure on a small first batch
        if (slots_per_task < 0 && (u64)nb * n_pos * generous > target_slots && !rk_knob("RK_BUILD_BATCH_NODES"))
            nb = std::max<u32>(1, (u32)std::min<u64>(nb, target_slots / ((u64)n_pos * generous)));
        a.task_base = (u64)node0 * n_pos;
        a.n_tasks = (u64)nb * n_pos;
        DevBuf b_keys, b_scores;
        u64 capacity = slots_per_task < 0 ? a.n_tasks * generous : (u64)(slots_per_task * 1.3 * (double)a.n_tasks);
        capacity = std::min<u64>(std::max<u64>(capacity, 1u << 20), target_slots > run_n ? target_slots - run_n : 1u << 20);
        capacity += chunk_slack;
        u64 counters[4] = {0, 0, 0, 0};  // tasks, slots, visits, tuples
        float explore_ms = 0.0f;
        for (int attempt = 0;; attempt++) {
            // room for the running set behind the chunks: it is sorted together with the new tuples
            if ((rc = b_keys.alloc((capacity + run_n) * 8)) || (rc = b_scores.alloc((capacity + run_n) * 4))) return rc;
            a.capacity = capacity; a.keys = b_keys.as<u64>(); a.scores = b_scores.as<float>();
            RK_HIP_TRY(hipMemset(b_cnt.p, 0, 4 * 8));
            RK_HIP_TRY(hipEventRecord(e0, 0));
            // (a shorter in-register tail does not help wide alphabets: measured 32-34 Gvisits/s for INL = 1, 2, 3 on AA k=5
            //  before the chunked append, which is what that case was waiting for)
            int inl = 3;
            if (const char *e = rk_knob("RK_BUILD_INLINE_LEVELS")) inl = atoi(e);  // developer knob
            const dim3 grid(blocks), block(64 * BUILD_WAVES_PER_BLOCK);
            // whole site rows in registers for the two alphabets RAPPAS has (4 / 20 states); anything else: per-visit loads
            const int vec = rk_knob("RK_BUILD_NO_VEC") ? 0 : (d->n_states == 4 ? 4 : (d->n_states == 20 ? 20 : 0));
#define RK_LAUNCH_EXPLORE(I)                                                                                \
    do {                                                                                                    \
        if (a.do_gap) {                                                                                     \
            if (vec == 4) hipLaunchKernelGGL((explore_kernel<I, 4, true>), grid, block, 0, 0, a);           \
            else if (vec == 20) hipLaunchKernelGGL((explore_kernel<I, 20, true>), grid, block, 0, 0, a);    \
            else hipLaunchKernelGGL((explore_kernel<I, 0, true>), grid, block, 0, 0, a);                    \
        } else if (vec == 4) hipLaunchKernelGGL((explore_kernel<I, 4, false>), grid, block, 0, 0, a);       \
        else if (vec == 20) hipLaunchKernelGGL((explore_kernel<I, 20, false>), grid, block, 0, 0, a);       \
        else hipLaunchKernelGGL((explore_kernel<I, 0, false>), grid, block, 0, 0, a);                       \
    } while (0)
            if (inl >= 3) RK_LAUNCH_EXPLORE(3);
            else if (inl == 2) RK_LAUNCH_EXPLORE(2);
            else RK_LAUNCH_EXPLORE(1);
#undef RK_LAUNCH_EXPLORE
            RK_HIP_TRY(hipGetLastError());
            RK_HIP_TRY(hipEventRecord(e1, 0));
            RK_HIP_TRY(hipEventSynchronize(e1));
            RK_HIP_TRY(hipEventElapsedTime(&explore_ms, e0, e1));
            RK_HIP_TRY(hipMemcpy(counters, b_cnt.p, sizeof(counters), hipMemcpyDeviceToHost));
            explore_total += explore_ms;
            if (counters[1] <= capacity) break;
            // the hand-out of explorers to waves is dynamic, so the padding of a second run can differ a little: leave slack
            if (attempt == 2) return fail_msg(RK_ERR_HIP, "rk_build_db: tuple buffer still too small after two resizes");
            capacity = counters[1] + chunk_slack;
            if (capacity + run_n >= (1ull << 31) - 1) break;  // too many for one sort: handled below
        }
        const u64 n_slots = counters[1];  // chunks handed out: real tuples + TUPLE_PAD fillers, every slot written
        const u64 n_tuples = counters[3];
        if (n_slots + run_n >= (1ull << 31) - 1 || n_slots > capacity) {
            if (nb == 1) return fail_msg(RK_ERR_UNSUPPORTED, "rk_build_db: node %u alone registers %llu tuples on top of %llu database entries: more than one sort can take", node0, (unsigned long long)n_tuples, (unsigned long long)run_n);
            batch = nb / 2;  // redo this range in smaller pieces
            continue;
        }
        total_tuples += n_tuples;
        total_visits += counters[2];
        // ---- sort by (code, branch), max per key ----
        RK_HIP_TRY(hipEventRecord(e1, 0));
        if (n_tuples + run_n) {
            if (run_n) {  // the running set goes behind the chunks; TUPLE_PAD fillers still sort last
                RK_HIP_TRY(hipMemcpyAsync(b_keys.as<u64>() + n_slots, run_keys.p, run_n * 8, hipMemcpyDeviceToDevice, 0));
                RK_HIP_TRY(hipMemcpyAsync(b_scores.as<float>() + n_slots, run_vals.p, run_n * 4, hipMemcpyDeviceToDevice, 0));
            }
            const int n_sort = (int)(n_slots + run_n);
            const int n = (int)(n_tuples + run_n);  // real pairs: the first n of the sorted slots
            DevBuf b_keys2, b_scores2, b_tmp, b_nruns;
            if ((rc = b_keys2.alloc((size_t)n_sort * 8)) || (rc = b_scores2.alloc((size_t)n_sort * 4)) || (rc = b_nruns.alloc(8))) return rc;
            // rocPRIM called natively (ROCm's own device primitives, no CUB-compatibility layer in between)
            rocprim::double_buffer<u64> kb(b_keys.as<u64>(), b_keys2.as<u64>());
            rocprim::double_buffer<float> vb(b_scores.as<float>(), b_scores2.as<float>());
            size_t tmp_bytes = 0;
            RK_HIP_TRY(rocprim::radix_sort_pairs(nullptr, tmp_bytes, kb, vb, (size_t)n_sort, 0u, (unsigned)end_bit, (hipStream_t)0));
            if ((rc = b_tmp.alloc(tmp_bytes))) return rc;
            RK_HIP_TRY(rocprim::radix_sort_pairs(b_tmp.p, tmp_bytes, kb, vb, (size_t)n_sort, 0u, (unsigned)end_bit, (hipStream_t)0));
            u64 *sorted_k = kb.current();
            float *sorted_v = vb.current();
            u64 *uniq_k = kb.alternate();
            float *max_v = vb.alternate();
            tmp_bytes = 0;
            RK_HIP_TRY(rocprim::reduce_by_key(nullptr, tmp_bytes, sorted_k, sorted_v, (size_t)n, uniq_k, max_v, b_nruns.as<int>(),
                                              rocprim::maximum<float>(), rocprim::equal_to<u64>(), (hipStream_t)0));
            if ((rc = b_tmp.alloc(tmp_bytes))) return rc;
            RK_HIP_TRY(rocprim::reduce_by_key(b_tmp.p, tmp_bytes, sorted_k, sorted_v, (size_t)n, uniq_k, max_v, b_nruns.as<int>(),
                                              rocprim::maximum<float>(), rocprim::equal_to<u64>(), (hipStream_t)0));
            int runs = 0;
            RK_HIP_TRY(hipMemcpy(&runs, b_nruns.p, 4, hipMemcpyDeviceToHost));
            if ((rc = run_keys.alloc((size_t)runs * 8)) || (rc = run_vals.alloc((size_t)runs * 4))) return rc;
            RK_HIP_TRY(hipMemcpy(run_keys.p, uniq_k, (size_t)runs * 8, hipMemcpyDeviceToDevice));
            RK_HIP_TRY(hipMemcpy(run_vals.p, max_v, (size_t)runs * 4, hipMemcpyDeviceToDevice));
            run_n = (u64)runs;
        }
        RK_HIP_TRY(hipEventRecord(e2, 0));
        RK_HIP_TRY(hipEventSynchronize(e2));
        float ms = 0.0f;
        RK_HIP_TRY(hipEventElapsedTime(&ms, e1, e2));
        reduce_total += ms;
        // next batch: aim at target_slots from what this one produced per node
        slots_per_task = std::max(1.0, (double)n_slots / (double)a.n_tasks);
        if (!rk_knob("RK_BUILD_BATCH_NODES")) {
            const u64 per_node = std::max<u64>(1, n_slots / nb);
            const u64 want = std::max<u64>(1, target_slots / per_node);
            batch = (u32)std::min<u64>(want, d->n_nodes);
        }
        node0 += nb;
    }
    out->tuples = total_tuples;
    out->visits = total_visits;
    out->explore_ms = explore_total;

    // ---- run lengths of the codes over the unique (code, branch) keys -> CSR, straight into the caller's arrays ----
    const u64 n_entries = run_n;
    out->n_entries = n_entries;
    out->branch_ids = (uint16_t *)malloc((n_entries + 1) * 2);
    out->scores = (float *)malloc((n_entries + 1) * 4);
    if (!out->branch_ids || !out->scores) { rk_built_free(out); return fail_msg(RK_ERR_NOMEM, "rk_build_db: host OOM for %llu entries", (unsigned long long)n_entries); }
    u64 n_keys = 0;
    std::vector<int> h_counts;
    if (run_n) {
        RK_HIP_TRY(hipEventRecord(e1, 0));
        const int runs = (int)run_n;
        DevBuf b_tmp, b_nruns, b_codes, b_counts, b_branch;
        if ((rc = b_codes.alloc((size_t)runs * 8)) || (rc = b_counts.alloc((size_t)runs * 4)) || (rc = b_nruns.alloc(8)) ||
            (rc = b_branch.alloc((size_t)runs * 2))) return rc;
        auto code_it = rocprim::make_transform_iterator(run_keys.as<u64>(), ShiftRight16());
        size_t tmp_bytes = 0;
        RK_HIP_TRY(rocprim::run_length_encode(nullptr, tmp_bytes, code_it, (unsigned)runs, b_codes.as<u64>(), b_counts.as<int>(), b_nruns.as<int>(), (hipStream_t)0));
        if ((rc = b_tmp.alloc(tmp_bytes))) return rc;
        RK_HIP_TRY(rocprim::run_length_encode(b_tmp.p, tmp_bytes, code_it, (unsigned)runs, b_codes.as<u64>(), b_counts.as<int>(), b_nruns.as<int>(), (hipStream_t)0));
        hipLaunchKernelGGL(split_keys_kernel, dim3((unsigned)prop.multiProcessorCount * 8), dim3(256), 0, 0, run_keys.as<u64>(), run_n, b_branch.as<unsigned short>());
        RK_HIP_TRY(hipGetLastError());
        int nk = 0;
        RK_HIP_TRY(hipMemcpy(&nk, b_nruns.p, 4, hipMemcpyDeviceToHost));
        n_keys = (u64)nk;
        RK_HIP_TRY(hipEventRecord(e2, 0));
        RK_HIP_TRY(hipEventSynchronize(e2));
        float ms = 0.0f;
        RK_HIP_TRY(hipEventElapsedTime(&ms, e1, e2));
        reduce_total += ms;
        out->key_codes = (uint64_t *)malloc((n_keys + 1) * 8);
        out->row_offsets = (uint64_t *)malloc((n_keys + 1) * 8);
        if (!out->key_codes || !out->row_offsets) { rk_built_free(out); return fail_msg(RK_ERR_NOMEM, "rk_build_db: host OOM for %llu keys", (unsigned long long)n_keys); }
        h_counts.resize(n_keys);
        RK_HIP_TRY(hipMemcpy(out->key_codes, b_codes.p, n_keys * 8, hipMemcpyDeviceToHost));
        RK_HIP_TRY(hipMemcpy(h_counts.data(), b_counts.p, n_keys * 4, hipMemcpyDeviceToHost));
        RK_HIP_TRY(hipMemcpy(out->branch_ids, b_branch.p, n_entries * 2, hipMemcpyDeviceToHost));
        RK_HIP_TRY(hipMemcpy(out->scores, run_vals.p, n_entries * 4, hipMemcpyDeviceToHost));
    } else {
        out->key_codes = (uint64_t *)malloc(8);
        out->row_offsets = (uint64_t *)malloc(8);
        if (!out->key_codes || !out->row_offsets) { rk_built_free(out); return fail_msg(RK_ERR_NOMEM, "rk_build_db: host OOM"); }
    }
    out->reduce_ms = reduce_total;
    out->n_keys = n_keys;
    u64 acc = 0;
    for (u64 i = 0; i < n_keys; i++) {
        out->row_offsets[i] = acc;
        acc += (u64)h_counts[i];
    }
    out->row_offsets[n_keys] = acc;
    return RK_OK;
}
