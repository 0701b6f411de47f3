// rk_synth_impl.h -- the seeded synthetic phylo-kmer database of SURVEY.md section 8(d), generated straight into the
// HBM image (included at the end of rk_engine.hip), and rk_db_fetch_row (a row read back out of the image).
//
// Why it lives in the library: BASELINE config C5 is a ~200 GB database (3.3e10 entries).  It cannot pass through
// rk_db_create's host arrays (the image plus its CSR source would not fit next to each other anywhere), so the image is
// filled by a kernel from a counter-based generator: every quantity is a pure function of (seed, dense k-mer index,
// entry index), integer arithmetic except one float32 multiply, so that a checker can regenerate any row on the host
// (rappas_amd/synth.py: synth_rows) and hand it to the oracle.
//
//   mix(x)      splitmix64 finaliser
//   h0          mix(seed + (dense + 1) * 0x9E3779B97F4A7C15)
//   present     (h0 >> 32) < floor(key_fraction * 2^32)
//   len         1 + #{ j in [1, max_len) : surv[j] > (mix(h0 ^ 0xA0761D6478BD642F) >> 32) },  max_len = max(1, n_branches - 1),
//               surv[0] = 2^32, surv[j] = (surv[j-1] * q) >> 32, q = floor((1 - 1/mean_row_len) * 2^32)   (1 + geometric)
//   b0          1 + (((mix(h0 ^ 0xE7037ED1A0B428DB) >> 32) * max(1, n_branches - len)) >> 32)   (0 when n_branches == 1)
//   entry i     branch b0 + i,  score = T * ((mix(h0 + (i + 1) * 0xD6E8FEB86659FD93) >> 40) * 2^-24)   (float32, T <= v <= 0)
#pragma once

namespace rk {

struct SynthParams {
    u64 seed, key_thresh;
    const u64 *surv;  // [surv_n], non-increasing
    u32 surv_n;
    u32 n_branches, max_len;
    float T;
};

__host__ __device__ __forceinline__ u64 synth_mix(u64 x) {
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 27; x *= 0x94D049BB133111EBull; x ^= x >> 31;
    return x;
}
__host__ __device__ __forceinline__ u64 synth_h0(u64 seed, u64 dense) { return synth_mix(seed + (dense + 1) * 0x9E3779B97F4A7C15ull); }
__host__ __device__ __forceinline__ u32 synth_len(const SynthParams &sp, u64 h0) {
    const u64 r = synth_mix(h0 ^ 0xA0761D6478BD642Full) >> 32;
    // first j in [1, surv_n) with surv[j] <= r (surv is non-increasing); all j below it count
    u32 lo = 1, hi = sp.surv_n;
    while (lo < hi) {
        const u32 mid = (lo + hi) >> 1;
        if (sp.surv[mid] > r) lo = mid + 1; else hi = mid;
    }
    const u32 len = lo;  // 1 + (lo - 1)
    return len < sp.max_len ? len : sp.max_len;
}
__host__ __device__ __forceinline__ u32 synth_b0(const SynthParams &sp, u64 h0, u32 len) {
    if (sp.n_branches == 1) return 0;
    const u64 hi = sp.n_branches > len ? sp.n_branches - len : 1;
    return 1u + (u32)(((synth_mix(h0 ^ 0xE7037ED1A0B428DBull) >> 32) * hi) >> 32);
}
__host__ __device__ __forceinline__ float synth_score(const SynthParams &sp, u64 h0, u32 i) {
    const u64 hs = synth_mix(h0 + (u64)(i + 1) * 0xD6E8FEB86659FD93ull);
    const float u = (float)(u32)(hs >> 40) * 5.9604644775390625e-08f;  // 24 random bits * 2^-24: exact
    return sp.T * u;
}

// one workgroup per row (grid-stride): INDEXED = the large-tree image (index line | u16 branch[lenp] | f32 score[lenp]),
// else 128-byte units of {slot offset, score} entries
template <bool INDEXED>
__global__ void __launch_bounds__(256) synth_fill_kernel(SynthParams sp, u64 n_keys, const u64 *dense, const u64 *desc,
                                                         unsigned char *rows) {
    for (u64 r = blockIdx.x; r < n_keys; r += gridDim.x) {
        const u64 h0 = synth_h0(sp.seed, dense[r]);
        const u32 len = synth_len(sp, h0);
        const u32 b0 = synth_b0(sp, h0, len);
        const u64 d = desc[r];
        const u32 lenp = (u32)d & DESC_LEN_MASK;
        unsigned char *row = rows + ((d >> DESC_LEN_BITS) << 3);
        if (INDEXED) {
            unsigned short *bp = (unsigned short *)row;
            float *scp = (float *)(row + 2 * (size_t)lenp);
            if (threadIdx.x < 32) {
                const u32 i = threadIdx.x + 1;
                const u32 bound = (u32)(((u64)i * sp.n_branches) / 32);
                const u32 below = bound > b0 ? bound - b0 : 0u;  // entries with branch < bound
                ((unsigned short *)(row - 64))[threadIdx.x] = (unsigned short)(below < len ? below : len);
            }
            for (u32 i = threadIdx.x; i < lenp; i += blockDim.x) {
                bp[i] = i < len ? (unsigned short)(b0 + i) : (unsigned short)0xFFFFu;
                scp[i] = i < len ? synth_score(sp, h0, i) : 0.0f;
            }
        } else {
            Entry *ep = (Entry *)row;
            for (u32 i = threadIdx.x; i < lenp; i += blockDim.x) {
                Entry e;
                e.branch = i < len ? (b0 + i + 1u) * 4u : 0u;
                e.score = i < len ? synth_score(sp, h0, i) : 0.0f;
                ep[i] = e;
            }
        }
    }
}

// one wave: the row of `code` as the placement kernels would see it (table lookup + entry decode)
template <int BITS, int TM>
__global__ void __launch_bounds__(64) fetch_row_kernel(DbView db, u64 code, u32 cap, u32 *out_len, unsigned short *out_br, float *out_sc) {
    const u32 lane = threadIdx.x;
    const u64 desc = lookup_desc<BITS, TM>(db, code);
    const u32 lenp = (u32)desc & DESC_LEN_MASK;
    u32 mine = 0;
    for (u32 e = lane; e < lenp; e += 64) {
        u32 br;
        float sc;
        load_entry(db, desc, e, br, sc);
        if (br == 0xFFFFu) continue;  // padding (only ever trails the row)
        mine++;
        if (e < cap) { out_br[e] = (unsigned short)br; out_sc[e] = sc; }
    }
    if (mine) atomicAdd(out_len, mine);
}

}  // namespace rk

namespace {
struct DevBufS {
    void *p = nullptr;
    ~DevBufS() { if (p) (void)hipFree(p); }
};
}  // namespace

extern "C" int rk_db_create_synth(const rk_synth_desc *d, rk_db **out) {
    RK_GUARD_BEGIN
    if (!d || !out) return fail(RK_ERR_INVALID, "rk_db_create_synth: null argument");
    *out = nullptr;
    if (d->alphabet != RK_ALPHABET_DNA && d->alphabet != RK_ALPHABET_AA)
        return fail(RK_ERR_INVALID, "rk_db_create_synth: alphabet must be 4 (DNA) or 20 (AA), got %u", d->alphabet);
    const uint32_t kmax = d->alphabet == RK_ALPHABET_DNA ? 31 : 12;
    if (d->k < 2 || d->k > kmax) return fail(RK_ERR_UNSUPPORTED, "rk_db_create_synth: k=%u outside supported range 2..%u", d->k, kmax);
    if (d->n_branches < 1 || d->n_branches > 65535) return fail(RK_ERR_INVALID, "rk_db_create_synth: n_branches=%u must be in 1..65535", d->n_branches);
    if (!std::isfinite(d->thr_log10) || !std::isfinite(d->thr)) return fail(RK_ERR_INVALID, "rk_db_create_synth: thresholds must be finite");
    if (!(d->key_fraction > 0.0 && d->key_fraction <= 1.0)) return fail(RK_ERR_INVALID, "rk_db_create_synth: key_fraction must be in (0, 1]");
    if (!(d->mean_row_len >= 1.0 && d->mean_row_len <= 65535.0)) return fail(RK_ERR_INVALID, "rk_db_create_synth: mean_row_len must be in [1, 65535]");
    uint64_t space = 0;
    if (!ipow_fits(d->alphabet, d->k, 1ull << 32, space))
        return fail(RK_ERR_UNSUPPORTED, "rk_db_create_synth: sigma^k must be <= 2^32 (every code is visited once)");
    uint32_t mode = d->table_mode;
    if (mode == RK_TABLE_AUTO) mode = space <= (1ull << 28) ? RK_TABLE_DIRECT : RK_TABLE_HASH;
    if ((mode == RK_TABLE_DIRECT || mode == RK_TABLE_DIRECT8) && space > (1ull << 31))
        return fail(RK_ERR_UNSUPPORTED, "rk_db_create_synth: direct table needs sigma^k <= 2^31 slots");
    if (mode != RK_TABLE_DIRECT && mode != RK_TABLE_DIRECT8 && mode != RK_TABLE_HASH)
        return fail(RK_ERR_INVALID, "rk_db_create_synth: bad table_mode %u", mode);

    // ---- generator tables ----
    SynthParams sp{};
    sp.seed = d->seed;
    sp.key_thresh = (uint64_t)(d->key_fraction * 4294967296.0);
    sp.n_branches = d->n_branches;
    sp.max_len = d->n_branches > 1 ? d->n_branches - 1 : 1;
    sp.T = d->thr_log10;
    std::vector<u64> surv;
    {
        const double p = 1.0 / d->mean_row_len;
        const uint64_t q = (uint64_t)((1.0 - p) * 4294967296.0);
        surv.push_back(1ull << 32);
        while (surv.size() < sp.max_len && surv.back() != 0) surv.push_back((surv.back() * q) >> 32);
    }
    sp.surv = surv.data();
    sp.surv_n = (uint32_t)surv.size();

    WindowPlan wp;
    const uint32_t sym_bits = d->alphabet == RK_ALPHABET_DNA ? 2u : 5u;
    // (used only when the image turns out not to be an indexed one; the rows are not drawn yet: their density from the spec)
    const double upc_spec = ((double)sp.key_thresh / 4294967296.0) * (d->mean_row_len / (double)ROW_UNIT + 0.5);
    const bool want_windows = window_plan(d->n_branches, sym_bits, upc_spec, d->mean_row_len / (double)ROW_UNIT + 0.5, wp);
    std::vector<unsigned char> winspec;  // [space] winspec_byte(first, last window) of every row (rows are branch runs)
    if (want_windows) {
        try { winspec.assign(space, 0); } catch (const std::bad_alloc &) { return fail(RK_ERR_NOMEM, "rk_db_create_synth: host OOM"); }
    }
    // ---- pass 1 (host): which codes carry a row and how long it is; rows are laid out in dense order ----
    std::vector<u64> dense, lens32;  // lens32: row length per present key (u64 to share the prefix pass below)
    {
        unsigned hw = std::thread::hardware_concurrency();
        const unsigned T = space > (1u << 20) ? std::max(1u, std::min(hw ? hw : 1u, 16u)) : 1u;
        std::vector<std::vector<u64>> dpart(T), lpart(T);
        std::vector<std::thread> th;
        bool oom = false;
        for (unsigned t = 0; t < T; t++) {
            auto job = [&, t]() {
                try {
                    const uint64_t lo = space * t / T, hi = space * (t + 1) / T;
                    for (uint64_t c = lo; c < hi; c++) {
                        const uint64_t h0 = synth_h0(sp.seed, c);
                        if ((h0 >> 32) >= sp.key_thresh) continue;
                        const u32 len = synth_len(sp, h0);
                        dpart[t].push_back(c);
                        lpart[t].push_back(len);
                        if (want_windows) {
                            const u32 b0 = synth_b0(sp, h0, len), f = b0 / wp.W, l = (b0 + len - 1) / wp.W;
                            winspec[c] = winspec_byte(f, l);
                        }
                    }
                } catch (const std::bad_alloc &) { oom = true; }
            };
            if (t + 1 < T) th.emplace_back(job); else job();
        }
        for (std::thread &x : th) x.join();
        if (oom) return fail(RK_ERR_NOMEM, "rk_db_create_synth: host OOM");
        try {
            for (unsigned t = 0; t < T; t++) {
                dense.insert(dense.end(), dpart[t].begin(), dpart[t].end());
                lens32.insert(lens32.end(), lpart[t].begin(), lpart[t].end());
                std::vector<u64>().swap(dpart[t]);
                std::vector<u64>().swap(lpart[t]);
            }
        } catch (const std::bad_alloc &) { return fail(RK_ERR_NOMEM, "rk_db_create_synth: host OOM"); }
    }
    const uint64_t n_keys = dense.size();
    uint64_t n_entries = 0;
    uint32_t max_len = 0;
    for (uint64_t i = 0; i < n_keys; i++) { n_entries += lens32[i]; if (lens32[i] > max_len) max_len = (uint32_t)lens32[i]; }
    const double mean_len = n_keys ? (double)n_entries / (double)n_keys : 0.0;
    uint64_t slot_units = 1;
    for (uint64_t i = 0; i < n_keys; i++) slot_units += (lens32[i] + ROW_UNIT - 1) / ROW_UNIT;
    const ImageKind kind = image_kind(d->n_branches, sym_bits, d->table_mode, space, true, slot_units, max_len, mean_len);
    const bool indexed = kind.indexed;
    const uint64_t unit_bytes = indexed ? 64 : ROW_UNIT * 8;
    uint64_t blob_units = 1, max_units = 0;
    std::vector<u64> &desc = lens32;  // lengths become descriptors in place
    for (uint64_t i = 0; i < n_keys; i++) {
        const uint64_t len = lens32[i];
        uint64_t units = (len + ROW_UNIT - 1) / ROW_UNIT, lenp = units * ROW_UNIT;
        if (indexed) {
            blob_units += 1;  // the index line
            lenp = (len + 31) / 32 * 32;
            units = lenp * 6 / 64;
        }
        if (units > max_units) max_units = units;
        desc[i] = ((blob_units * (unit_bytes / 8)) << DESC_LEN_BITS) | lenp;
        blob_units += units;
    }
    const uint64_t blob_bytes = blob_units * unit_bytes;
    if ((blob_bytes >> 3) >= (1ull << 40)) return fail(RK_ERR_UNSUPPORTED, "rk_db_create_synth: row blob exceeds 8 TiB");
    std::vector<uint64_t> table;
    uint64_t slots = 0, hash_mask = 0;
    bool nib = false;
    {
        const uint32_t alphabet = d->alphabet, k = d->k;
        auto code_of = [&](uint64_t i) -> uint64_t {
            if (alphabet == RK_ALPHABET_DNA) return dense[i];
            uint64_t code = 0, rem = dense[i];
            for (uint32_t j = 0; j < k; j++) { code |= (rem % 20) << (5 * j); rem /= 20; }
            return code;
        };
        int rc = build_table(mode, space, n_keys, indexed, max_units, blob_units, [&](uint64_t i) { return dense[i]; },
                             [&](uint64_t i) { return desc[i]; }, code_of, table, slots, hash_mask, nib);
        if (rc) return rc;
    }

    // ---- device: table upload, fill kernel ----
    rk_db *db = nullptr;
    int prev = 0;
    (void)hipGetDevice(&prev);
    struct Restore { int p; ~Restore() { (void)hipSetDevice(p); } } restore{prev};
    DbMeta meta{d->alphabet, d->convert_uo, d->k, d->n_branches, d->thr_log10, d->thr};
    int rc = open_db(meta, d->device, &db);
    if (rc) return rc;
    DevBufS d_dense, d_desc, d_surv;
#define SY_TRY(expr)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess) {                                                                   \
            int c_ = fail(e_ == hipErrorOutOfMemory ? RK_ERR_NOMEM : RK_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
            rk_db_destroy(db);                                                                    \
            return c_;                                                                            \
        }                                                                                         \
    } while (0)
    const size_t table_bytes = table.size() * sizeof(uint64_t);
    SY_TRY(hipMalloc(&db->d_table, table_bytes ? table_bytes : 8));
    SY_TRY(hipMalloc(&db->d_rows, blob_bytes));
    if (table_bytes) SY_TRY(hipMemcpy(db->d_table, table.data(), table_bytes, hipMemcpyHostToDevice));
    SY_TRY(hipMemset(db->d_rows, indexed ? 0xFF : 0, unit_bytes));  // unit 0: the reserved "skip" / scratch pattern
    if (want_windows && kind.windowable && mode == RK_TABLE_DIRECT && blob_bytes < RK_WINDOW_MAX_BLOB) {
        SY_TRY(hipMalloc((void **)&db->d_winspec, winspec.size()));
        SY_TRY(hipMemcpy(db->d_winspec, winspec.data(), winspec.size(), hipMemcpyHostToDevice));
        db->windowed = true;
        db->wp = wp;
    }
    if (n_keys) {
        SY_TRY(hipMalloc(&d_dense.p, n_keys * 8));
        SY_TRY(hipMalloc(&d_desc.p, n_keys * 8));
        SY_TRY(hipMalloc(&d_surv.p, surv.size() * 8));
        SY_TRY(hipMemcpy(d_dense.p, dense.data(), n_keys * 8, hipMemcpyHostToDevice));
        SY_TRY(hipMemcpy(d_desc.p, desc.data(), n_keys * 8, hipMemcpyHostToDevice));
        SY_TRY(hipMemcpy(d_surv.p, surv.data(), surv.size() * 8, hipMemcpyHostToDevice));
        SynthParams dsp = sp;
        dsp.surv = (const u64 *)d_surv.p;
        uint64_t blocks = (uint64_t)db->cu_count * 16;
        if (blocks > n_keys) blocks = n_keys;
        if (indexed)
            hipLaunchKernelGGL(synth_fill_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, 0, dsp, (u64)n_keys, (const u64 *)d_dense.p,
                               (const u64 *)d_desc.p, (unsigned char *)db->d_rows);
        else
            hipLaunchKernelGGL(synth_fill_kernel<false>, dim3((unsigned)blocks), dim3(64), 0, 0, dsp, (u64)n_keys, (const u64 *)d_dense.p,
                               (const u64 *)d_desc.p, (unsigned char *)db->d_rows);
        SY_TRY(hipGetLastError());
        SY_TRY(hipDeviceSynchronize());
    }
#undef SY_TRY
    db->compact_nib = nib;
    finish_db(db, meta, mode, indexed, /*mono=*/d->thr_log10 <= 0.0f, n_keys, n_entries, slots, hash_mask, table_bytes, blob_bytes, max_len);
    rc = check_launchable(db);
    if (rc) { rk_db_destroy(db); return rc; }
    *out = db;
    return RK_OK;
    RK_GUARD_END("rk_db_create_synth")
}

template <int BITS>
static void launch_fetch(const rk_db *db, u64 code, u32 cap, u32 *d_len, unsigned short *d_br, float *d_sc) {
    switch (db->info.table_mode) {
    case RK_TABLE_DIRECT: hipLaunchKernelGGL((fetch_row_kernel<BITS, TM_COMPACT>), dim3(1), dim3(64), 0, 0, db->view, code, cap, d_len, d_br, d_sc); break;
    case RK_TABLE_DIRECT8: hipLaunchKernelGGL((fetch_row_kernel<BITS, TM_DIRECT8>), dim3(1), dim3(64), 0, 0, db->view, code, cap, d_len, d_br, d_sc); break;
    default: hipLaunchKernelGGL((fetch_row_kernel<BITS, TM_HASH>), dim3(1), dim3(64), 0, 0, db->view, code, cap, d_len, d_br, d_sc); break;
    }
}

extern "C" int rk_db_fetch_row(rk_db *db, uint64_t code, uint32_t cap, uint32_t *len, uint16_t *branch_ids, float *scores) {
    if (!db || !len) return fail(RK_ERR_INVALID, "rk_db_fetch_row: null argument");
    if (cap && (!branch_ids || !scores)) return fail(RK_ERR_INVALID, "rk_db_fetch_row: null output arrays");
    const uint32_t bits = db->info.bits_per_symbol;
    if (bits * db->info.k < 64 && (code >> (bits * db->info.k))) return fail(RK_ERR_INVALID, "rk_db_fetch_row: code outside the k-mer space");
    if (bits == 5)
        for (uint32_t i = 0; i < db->info.k; i++)
            if (((code >> (5 * i)) & 31) >= 20) return fail(RK_ERR_INVALID, "rk_db_fetch_row: code has a digit >= 20");
    int prev = 0;
    (void)hipGetDevice(&prev);
    struct Restore { int p; ~Restore() { (void)hipSetDevice(p); } } restore{prev};
    HIP_TRY(hipSetDevice(db->info.device));
    DevBufS d_len, d_br, d_sc;
    HIP_TRY(hipMalloc(&d_len.p, 4));
    HIP_TRY(hipMalloc(&d_br.p, (size_t)(cap ? cap : 1) * 2));
    HIP_TRY(hipMalloc(&d_sc.p, (size_t)(cap ? cap : 1) * 4));
    HIP_TRY(hipMemset(d_len.p, 0, 4));
    if (bits == 2) launch_fetch<2>(db, code, cap, (u32 *)d_len.p, (unsigned short *)d_br.p, (float *)d_sc.p);
    else launch_fetch<5>(db, code, cap, (u32 *)d_len.p, (unsigned short *)d_br.p, (float *)d_sc.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(len, d_len.p, 4, hipMemcpyDeviceToHost));
    const uint32_t n = *len < cap ? *len : cap;
    if (n) {
        HIP_TRY(hipMemcpy(branch_ids, d_br.p, (size_t)n * 2, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(scores, d_sc.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    }
    return RK_OK;
}
