// rk_device.h -- shared device-side definitions for the gfx950 placement kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rk {

typedef unsigned long long u64;
typedef unsigned int u32;

// Row descriptor (8 B): (offset of the row in the rows blob, in 8-byte units) << 24 | row length.
// len == 0 <=> k-mer absent.  40-bit offsets address 8 TiB of rows; rows hold < 2^24 entries.
constexpr int DESC_LEN_BITS = 24;
constexpr u32 DESC_LEN_MASK = (1u << DESC_LEN_BITS) - 1;

// Row image in the blob: len entries of 8 bytes {u32 slot offset, f32 score}, 8-byte aligned.
// One dwordx2 load per lane fetches an entry; chunk c of a row simply starts G entries further.
// `branch` holds the BYTE offset of the branch's word in the per-read score vector, (branch id + 1) * 4: word 0 of
// the vector is a scratch slot, so all-zero padding entries (and whatever an out-of-range buffer load returns: zeros)
// update the scratch word and need no test in the accumulate loop.  (Large-tree images keep raw u16 ids, see soa.)
// (Algorithmic size of an entry is 6 bytes -- u16 + f32 -- that is what the roofline accounting uses.)
struct __attribute__((aligned(8))) Entry {
    u32 branch;
    float score;
};
#ifndef RK_FIT32_LIMIT
#define RK_FIT32_LIMIT 0xFFFF0000u  // (developer builds lower it to time the 64-bit-offset kernels on small databases)
#endif
constexpr u32 ROWS_FIT32_LIMIT = RK_FIT32_LIMIT;  // blobs below this use 32-bit offsets (ITEM_FILLER must stay outside the buffer)
constexpr u32 ROW_UNIT = 16;  // entries per 128-byte unit; rows are unit-aligned and padded to whole units

// k-mer -> row lookup flavours (template parameter of the kernels)
constexpr int TM_HASH = 0, TM_DIRECT8 = 1, TM_COMPACT = 2;
constexpr u32 COMPACT_KMERS = 12;  // k-mers per 16-byte block of the compact direct table

// Untouched marker of the per-read LDS score vector: -inf.  No finite sum of finite scores produces it, it loses every
// '>' comparison of the select phase, and max(marker, Q*T) seeds a first touch.
constexpr u32 S_UNTOUCHED = 0xFF800000u;

struct DbView {
    const u64 *direct;   // [sigma^k] row descriptors (RK_TABLE_DIRECT8) or nullptr
    const uint4 *compact;  // RK_TABLE_DIRECT: [ceil(sigma^k/12)] blocks {u32 first unit, 12 x u8 units}, or -- compact_nib -- [ceil(sigma^k/24)] blocks
                           // {u32 first unit, 24 x u4 units} when no row exceeds 15 units; nullptr otherwise
    const uint4 *slots;  // [hash_mask+1] {key+1 lo, key+1 hi, desc lo, desc hi} (RK_TABLE_HASH) or nullptr
    u64 hash_mask;
    const unsigned char *rows;
    u64 rows_bytes;
    u32 k, bits, n_branches, alphabet;
    float T, P;
    u32 convert_uo;
    u32 compact_nib;  // the compact table holds 4-bit unit counts, 24 k-mers per block (half the table: its lines stay in the L2)
    u32 mono;  // every score >= T (all increments >= 0): first touch of a branch can be a max with the -inf marker
    u32 soa;  // large-tree (indexed) images: a row is u16 branch[len] followed by f32 score[len] (6 bytes per entry)
    // mid-size trees (place_packed16w_kernel): the tree is cut into n_win windows of win_w branches; winspec[dense k-mer index] =
    // (first window its row touches, 6 bits) | min(last - first, 3) << 6 (3 = to the last window), 0 for absent k-mers; nullptr when
    // the image is not windowed
    const unsigned char *winspec;
    u32 win_w, n_win;
};

struct PlaceArgs {
    DbView db;
    u64 n_reads;
    const u32 *packed;
    u32 words_per_read;
    const u32 *lens;
    u32 fixed_len;
    const u32 *flags_in;
    u32 has_ascii;  // ambiguous reads are handled by the ASCII kernel; leave their outputs alone
    // params
    u32 keep_at_most;
    float keep_factor;
    float ns_bound;
    // outputs
    unsigned char *o_nrows;
    unsigned short *o_branch;
    float *o_score;
    double *o_lwr;
    u32 *o_flags;
    // geometry
    u32 s_stride;  // u32 words per read score vector in LDS
    u32 list_cap;  // u64 slots of the per-read hit list in LDS
    u32 n_pass;    // large-tree kernels: branch-range passes per read (1 unless the score vector exceeds one CU's LDS)
    u32 main_cap, work_cap;  // windowed kernels: u32 slots of the per-read item list / of the per-window work (touched-slot) list
    u32 only_marked;         // place_packed16w_kernel as the second launch: only the tiles marked in tile_marks
    u32 only_if;             // first kernels launched side by side: 0 = run; else the classes of batches this launch is for, as the pre-pass judged the
                             // batch on the device -- bit 0: uniform reads (*keep_order != 0) whose sampled k-mers have a row more often than a random
                             // read's, bit 1: uniform reads that hit no more often than that (keep_order[3] != 0), bit 2: reads of a clade (re-tiled)
    unsigned char *tile_marks;  // [ceil(n_reads / 4)] scratch of the launch (zeroed before the first kernel): tile t -- the reads at slots 4t .. 4t+3 of the
                                // batch's order -- is left to place_packed16w_kernel by the kernel launched ahead of it (never the caller's flag array:
                                // d_flags_in may be the same buffer as the output flags)
    const u32 *perm;         // tile t holds reads perm[4t .. 4t+3] (reads grouped by their place in the tree); null = in order
    u32 *marked_list;        // [<= n_tiles] the marked tiles as a list (compact_marks_kernel) for the second launch, or nullptr: it scans tile_marks
    u32 *marked_ctl;         // {number of tiles in marked_list, head of the queue the second launch's waves take them from}, zeroed with the marks
    const u32 *keep_order;   // with perm: *keep_order != 0 = the pre-pass found a batch of reads without a clade and left perm unwritten
};

__device__ __forceinline__ u64 mix64(u64 x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return x;
}

}  // namespace rk
