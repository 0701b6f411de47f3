// rk_image_impl.h -- the persisted form of the HBM image (included by rk_engine.hip: it needs rk_db's insides).
//
// The reference stores and reloads its database as a Java-serialised session (src/main_v2/SessionNext_v2.java:110-154 store,
// :158-207 load; README: "JDK-version fragile").  Here the lookup structure is written exactly as it sits in HBM -- k-mer table,
// row blob, window spans -- behind a fixed header, so that loading is mmap + one host-to-device copy per section: no parse, no
// rebuild (SURVEY.md section 5 "checkpoint" row, section 7 step 3).  A caller-defined blob travels with it (rk_place keeps the
// tree there); the engine never looks inside.
//
// File layout (little-endian; sections start on 4 096-byte boundaries):
//   [0, 4096)   ImageHeader (below), zero-padded
//   table       info.table_bytes
//   rows        info.rows_bytes
//   winspec     winspec_bytes (sigma^k, images that carry window spans / position keys)
//   user        user_bytes
// header.payload_hash covers the four sections in file order, header.header_hash the header's bytes before it.
#pragma once
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace {
constexpr uint64_t RK_IMAGE_MAGIC = 0x31474D4942444B52ull;  // "RKDBIMG1"
constexpr uint32_t RK_IMAGE_VERSION = 1, RK_IMAGE_ALIGN = 4096;

struct ImageHeader {
    uint64_t magic;
    uint32_t version, header_bytes, endian_tag /* 0x01020304 */, engine_version;
    // the scalars of rk_db_desc / rk_db_info
    uint32_t alphabet, convert_uo, k, n_branches;
    float thr_log10, thr;
    uint32_t table_mode, bits_per_symbol, max_row_len;
    uint32_t indexed, mono, compact_nib, windowed, has_pos;
    uint64_t n_keys, n_entries, table_slots, hash_mask;
    // the windowed kernels' plan (WindowPlan)
    uint32_t wp_W, wp_n_win, wp_s_stride, wp_main_cap, wp_work_cap, wp_stream;
    double wp_units_per_code;
    // sections
    uint64_t table_bytes, rows_bytes, winspec_bytes, user_bytes;
    uint64_t payload_hash;
    uint64_t header_hash;  // of every byte above
};
static_assert(sizeof(ImageHeader) <= RK_IMAGE_ALIGN, "header fits its page");

inline uint64_t align_up(uint64_t v) { return (v + RK_IMAGE_ALIGN - 1) & ~(uint64_t)(RK_IMAGE_ALIGN - 1); }

// 64-bit checksum, streamed: eight interleaved multiply-xor lanes over 8-byte words (memory-bound on one core), tail bytes folded
// in at the end.  Not cryptographic: it is there to refuse files that were cut short, overwritten or bit-flipped.
struct Hash64 {
    uint64_t lane[8], total = 0;
    unsigned char tail[64];
    size_t n_tail = 0;
    Hash64() { for (int i = 0; i < 8; i++) lane[i] = 0x9E3779B97F4A7C15ull * (uint64_t)(i + 1); }
    void block(const unsigned char *p) {
        uint64_t w[8];
        memcpy(w, p, 64);
        for (int i = 0; i < 8; i++) {
            uint64_t h = (lane[i] ^ w[i]) * 0xD6E8FEB86659FD93ull;
            lane[i] = h ^ (h >> 29);
        }
    }
    void add(const void *data, size_t n) {
        const unsigned char *p = (const unsigned char *)data;
        total += n;
        if (n_tail) {
            const size_t take = std::min(n, sizeof(tail) - n_tail);
            memcpy(tail + n_tail, p, take);
            n_tail += take; p += take; n -= take;
            if (n_tail < sizeof(tail)) return;
            block(tail);
            n_tail = 0;
        }
        for (; n >= 64; n -= 64, p += 64) block(p);
        if (n) { memcpy(tail, p, n); n_tail = n; }
    }
    uint64_t digest() const {
        Hash64 c = *this;
        if (c.n_tail) { memset(c.tail + c.n_tail, 0, sizeof(c.tail) - c.n_tail); c.block(c.tail); }
        uint64_t h = c.total * 0x9E3779B97F4A7C15ull;
        for (int i = 0; i < 8; i++) { h = (h ^ c.lane[i]) * 0xD6E8FEB86659FD93ull; h ^= h >> 32; }
        return h;
    }
};

struct FileCloser {
    int fd = -1;
    ~FileCloser() { if (fd >= 0) (void)close(fd); }
};

int write_all(int fd, const void *data, size_t n, const char *path) {
    const unsigned char *p = (const unsigned char *)data;
    while (n) {
        const ssize_t w = write(fd, p, n > (1u << 30) ? (1u << 30) : n);
        if (w < 0) { if (errno == EINTR) continue; return fail(RK_ERR_IO, "rk_db_save: write to %s failed: %s", path, strerror(errno)); }
        p += w; n -= (size_t)w;
    }
    return RK_OK;
}
int pad_to_page(int fd, uint64_t written, const char *path) {
    static const unsigned char zeros[RK_IMAGE_ALIGN] = {0};
    const uint64_t pad = align_up(written) - written;
    return pad ? write_all(fd, zeros, (size_t)pad, path) : RK_OK;
}

// One section of an image that is being written: host memory, or device memory read back through a bounce buffer.
struct Section {
    const void *host = nullptr, *dev = nullptr;
    uint64_t bytes = 0;
};

int write_image(const char *path, ImageHeader h, const Section (&sec)[4], int device) {
    if (!path || !*path) return fail(RK_ERR_INVALID, "rk_db_save: empty path");
    const std::string tmp = std::string(path) + ".tmp";
    FileCloser f;
    f.fd = open(tmp.c_str(), O_WRONLY | O_CREAT | O_TRUNC | O_CLOEXEC, 0644);
    if (f.fd < 0) return fail(RK_ERR_IO, "rk_db_save: cannot create %s: %s", tmp.c_str(), strerror(errno));
    struct Unlink { std::string p; bool armed = true; ~Unlink() { if (armed) (void)unlink(p.c_str()); } } cleanup{tmp};
    // the header's page is written last (it carries the payload's checksum)
    static const unsigned char zeros[RK_IMAGE_ALIGN] = {0};
    if (int rc = write_all(f.fd, zeros, RK_IMAGE_ALIGN, path)) return rc;
    Hash64 hs;
    std::vector<unsigned char> bounce;
    for (const Section &s : sec) {
        if (s.host) {
            if (int rc = write_all(f.fd, s.host, (size_t)s.bytes, path)) return rc;
            hs.add(s.host, (size_t)s.bytes);
        } else if (s.bytes) {
            const size_t chunk = 64u << 20;
            if (bounce.empty()) bounce.resize((size_t)std::min<uint64_t>(chunk, std::max<uint64_t>({sec[0].bytes, sec[1].bytes, sec[2].bytes, 1})));
            HIP_TRY(hipSetDevice(device));
            for (uint64_t off = 0; off < s.bytes; off += bounce.size()) {
                const size_t n = (size_t)std::min<uint64_t>(bounce.size(), s.bytes - off);
                HIP_TRY(hipMemcpy(bounce.data(), (const unsigned char *)s.dev + off, n, hipMemcpyDeviceToHost));
                if (int rc = write_all(f.fd, bounce.data(), n, path)) return rc;
                hs.add(bounce.data(), n);
            }
        }
        if (int rc = pad_to_page(f.fd, s.bytes, path)) return rc;
    }
    h.payload_hash = hs.digest();
    Hash64 hh;
    hh.add(&h, offsetof(ImageHeader, header_hash));
    h.header_hash = hh.digest();
    if (lseek(f.fd, 0, SEEK_SET) != 0) return fail(RK_ERR_IO, "rk_db_save: seek in %s failed: %s", tmp.c_str(), strerror(errno));
    if (int rc = write_all(f.fd, &h, sizeof(h), path)) return rc;
    if (fsync(f.fd) != 0) return fail(RK_ERR_IO, "rk_db_save: fsync of %s failed: %s", tmp.c_str(), strerror(errno));
    (void)close(f.fd);
    f.fd = -1;
    if (rename(tmp.c_str(), path) != 0) return fail(RK_ERR_IO, "rk_db_save: cannot rename %s to %s: %s", tmp.c_str(), path, strerror(errno));
    cleanup.armed = false;
    return RK_OK;
}

void header_common(ImageHeader &h, uint32_t alphabet, uint32_t convert_uo, uint32_t k, uint32_t n_branches, float thr_log10, float thr) {
    memset(&h, 0, sizeof(h));
    h.magic = RK_IMAGE_MAGIC; h.version = RK_IMAGE_VERSION; h.header_bytes = (uint32_t)sizeof(ImageHeader); h.endian_tag = 0x01020304u;
    h.engine_version = RK_VERSION;
    h.alphabet = alphabet; h.convert_uo = convert_uo; h.k = k; h.n_branches = n_branches; h.thr_log10 = thr_log10; h.thr = thr;
}
void header_plan(ImageHeader &h, const WindowPlan &wp) {
    h.wp_W = wp.W; h.wp_n_win = wp.n_win; h.wp_s_stride = wp.s_stride; h.wp_main_cap = wp.main_cap; h.wp_work_cap = wp.work_cap;
    h.wp_stream = wp.stream ? 1u : 0u; h.wp_units_per_code = wp.units_per_code;
}

// A mapped image file whose header, size and (optionally) payload checksum have been checked.
struct MappedImage {
    const unsigned char *base = nullptr;
    size_t size = 0;
    ImageHeader h{};
    uint64_t off_table = 0, off_rows = 0, off_winspec = 0, off_user = 0;
    ~MappedImage() { if (base) (void)munmap((void *)base, size); }
    int open_file(const char *path, bool verify_payload, const char *who) {
        if (!path || !*path) return fail(RK_ERR_INVALID, "%s: empty path", who);
        FileCloser f;
        f.fd = open(path, O_RDONLY | O_CLOEXEC);
        if (f.fd < 0) return fail(RK_ERR_IO, "%s: cannot open %s: %s", who, path, strerror(errno));
        struct stat st;
        if (fstat(f.fd, &st) != 0) return fail(RK_ERR_IO, "%s: cannot stat %s: %s", who, path, strerror(errno));
        if ((uint64_t)st.st_size < RK_IMAGE_ALIGN) return fail(RK_ERR_IO, "%s: %s is not a database image (%lld bytes)", who, path, (long long)st.st_size);
        size = (size_t)st.st_size;
        void *m = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, f.fd, 0);
        if (m == MAP_FAILED) { size = 0; return fail(RK_ERR_NOMEM, "%s: cannot map %s: %s", who, path, strerror(errno)); }
        base = (const unsigned char *)m;
        memcpy(&h, base, sizeof(h));
        if (h.magic != RK_IMAGE_MAGIC) return fail(RK_ERR_IO, "%s: %s is not a database image (bad magic)", who, path);
        if (h.endian_tag != 0x01020304u) return fail(RK_ERR_UNSUPPORTED, "%s: %s was written on a machine of the other byte order", who, path);
        if (h.version != RK_IMAGE_VERSION || h.header_bytes != sizeof(ImageHeader))
            return fail(RK_ERR_UNSUPPORTED, "%s: %s has image version %u (header of %u bytes); this library reads version %u", who, path, h.version, h.header_bytes, RK_IMAGE_VERSION);
        Hash64 hh;
        hh.add(&h, offsetof(ImageHeader, header_hash));
        if (hh.digest() != h.header_hash) return fail(RK_ERR_IO, "%s: %s: header checksum mismatch (corrupted file)", who, path);
        off_table = RK_IMAGE_ALIGN;
        off_rows = off_table + align_up(h.table_bytes);
        off_winspec = off_rows + align_up(h.rows_bytes);
        off_user = off_winspec + align_up(h.winspec_bytes);
        const uint64_t need = off_user + align_up(h.user_bytes);
        if (h.table_bytes > (1ull << 46) || h.rows_bytes > (1ull << 46) || h.winspec_bytes > (1ull << 40) || h.user_bytes > (1ull << 40) || need != (uint64_t)size)
            return fail(RK_ERR_IO, "%s: %s is %llu bytes, its header describes %llu (truncated or overwritten file)", who, path,
                        (unsigned long long)size, (unsigned long long)need);
        // header sanity: what rk_db_create would have refused
        if ((h.alphabet != RK_ALPHABET_DNA && h.alphabet != RK_ALPHABET_AA) || h.n_branches < 1 || h.n_branches > 65535 || h.k < 2 || h.k > 31 ||
            h.bits_per_symbol != (h.alphabet == RK_ALPHABET_DNA ? 2u : 5u) ||
            (h.table_mode != RK_TABLE_HASH && h.table_mode != RK_TABLE_DIRECT && h.table_mode != RK_TABLE_DIRECT8) || h.rows_bytes < 128 || (h.rows_bytes & 7))
            return fail(RK_ERR_IO, "%s: %s: inconsistent header", who, path);
        if (h.table_mode == RK_TABLE_HASH && (h.table_slots == 0 || (h.table_slots & (h.table_slots - 1)) || h.hash_mask != h.table_slots - 1 || h.table_bytes != h.table_slots * 16))
            return fail(RK_ERR_IO, "%s: %s: inconsistent hash table geometry", who, path);
        {
            uint64_t space = 0;
            const bool fits = ipow_fits(h.alphabet, h.k, 1ull << 40, space);
            if ((h.windowed || h.has_pos) && (!fits || h.winspec_bytes != space)) return fail(RK_ERR_IO, "%s: %s: window spans missing or of the wrong size", who, path);
            if (h.table_mode == RK_TABLE_DIRECT8 && (!fits || h.table_bytes != space * 8)) return fail(RK_ERR_IO, "%s: %s: inconsistent table size", who, path);
            if (h.table_mode == RK_TABLE_DIRECT) {
                const uint64_t per = h.compact_nib ? 2 * COMPACT_KMERS : COMPACT_KMERS;
                if (!fits || h.table_bytes != (space + per - 1) / per * 16) return fail(RK_ERR_IO, "%s: %s: inconsistent table size", who, path);
            }
            if (h.windowed && (h.wp_n_win < 1 || h.wp_n_win > RK_MAX_WINDOWS || h.wp_W < 4 || (h.wp_W & 3) || (uint64_t)h.wp_W * h.wp_n_win < h.n_branches || h.wp_W > 4096 ||
                               h.wp_s_stride != h.wp_W + 4 || h.wp_main_cap < 160 || h.wp_main_cap > 640 || (h.wp_main_cap & 1) || h.wp_work_cap > 200 || (h.wp_work_cap & 1) ||
                               h.table_mode != RK_TABLE_DIRECT || h.indexed || h.rows_bytes >= RK_WINDOW_MAX_BLOB))
                return fail(RK_ERR_IO, "%s: %s: inconsistent window plan", who, path);
            if (h.indexed && h.table_mode == RK_TABLE_DIRECT) return fail(RK_ERR_IO, "%s: %s: inconsistent header", who, path);
        }
        if (verify_payload) {
            (void)madvise((void *)base, size, MADV_SEQUENTIAL);
            Hash64 hs;
            hs.add(base + off_table, (size_t)h.table_bytes);
            hs.add(base + off_rows, (size_t)h.rows_bytes);
            hs.add(base + off_winspec, (size_t)h.winspec_bytes);
            hs.add(base + off_user, (size_t)h.user_bytes);
            if (hs.digest() != h.payload_hash) return fail(RK_ERR_IO, "%s: %s: payload checksum mismatch (corrupted file)", who, path);
        }
        return RK_OK;
    }
};

void info_from_header(const ImageHeader &h, rk_db_info *info, int device) {
    memset(info, 0, sizeof(*info));
    info->alphabet = h.alphabet; info->k = h.k; info->n_branches = h.n_branches; info->table_mode = h.table_mode;
    info->thr_log10 = h.thr_log10; info->thr = h.thr; info->n_keys = h.n_keys; info->n_entries = h.n_entries;
    info->table_slots = h.table_slots; info->table_bytes = h.table_bytes; info->rows_bytes = h.rows_bytes;
    info->bits_per_symbol = h.bits_per_symbol; info->max_row_len = h.max_row_len; info->device = device;
}
}  // namespace

extern "C" int rk_db_save(const rk_db *db, const char *path, const void *user, uint64_t user_bytes) {
    RK_GUARD_BEGIN
    if (!db) return fail(RK_ERR_INVALID, "rk_db_save: null handle");
    if (user_bytes && !user) return fail(RK_ERR_INVALID, "rk_db_save: user_bytes without a user blob");
    ImageHeader h;
    header_common(h, db->info.alphabet, db->convert_uo, db->info.k, db->info.n_branches, db->info.thr_log10, db->info.thr);
    h.table_mode = db->info.table_mode; h.bits_per_symbol = db->info.bits_per_symbol; h.max_row_len = db->info.max_row_len;
    h.indexed = db->indexed; h.mono = db->view.mono; h.compact_nib = db->compact_nib; h.windowed = db->windowed; h.has_pos = db->has_pos;
    h.n_keys = db->info.n_keys; h.n_entries = db->info.n_entries; h.table_slots = db->info.table_slots; h.hash_mask = db->view.hash_mask;
    header_plan(h, db->wp);
    uint64_t space = 0;
    if (db->windowed || db->has_pos) (void)ipow_fits(db->info.alphabet, db->info.k, 1ull << 40, space);
    h.table_bytes = db->info.table_bytes; h.rows_bytes = db->info.rows_bytes; h.winspec_bytes = space; h.user_bytes = user_bytes;
    Section sec[4];
    sec[0].dev = db->d_table; sec[0].bytes = h.table_bytes;
    sec[1].dev = db->d_rows; sec[1].bytes = h.rows_bytes;
    sec[2].dev = db->d_winspec; sec[2].bytes = h.winspec_bytes;
    sec[3].host = user_bytes ? user : (const void *)""; sec[3].bytes = user_bytes;
    int prev = 0;
    (void)hipGetDevice(&prev);
    struct Restore { int p; ~Restore() { (void)hipSetDevice(p); } } restore{prev};
    HIP_TRY(hipSetDevice(db->info.device));
    HIP_TRY(hipDeviceSynchronize());
    return write_image(path, h, sec, db->info.device);
    RK_GUARD_END("rk_db_save")
}

// The same file from the caller's CSR arrays, built on the host: no device is touched (a build machine without a GPU can write
// the image a placement node will load).  desc->device is ignored.
extern "C" int rk_db_save_desc(const rk_db_desc *d, const char *path, const void *user, uint64_t user_bytes) {
    RK_GUARD_BEGIN
    if (!d) return fail(RK_ERR_INVALID, "rk_db_save_desc: null argument");
    if (user_bytes && !user) return fail(RK_ERR_INVALID, "rk_db_save_desc: user_bytes without a user blob");
    DbImage img;
    if (int rc = build_image(d, img)) return rc;
    ImageHeader h;
    header_common(h, d->alphabet, d->convert_uo, d->k, d->n_branches, d->thr_log10, d->thr);
    h.table_mode = img.mode; h.bits_per_symbol = img.bits; h.max_row_len = img.max_len;
    h.indexed = img.indexed; h.mono = img.mono; h.compact_nib = img.nib; h.windowed = img.windowed; h.has_pos = img.has_pos;
    h.n_keys = img.n_keys; h.n_entries = img.n_entries; h.table_slots = img.slots; h.hash_mask = img.hash_mask;
    if (img.windowed) header_plan(h, img.wp);
    h.table_bytes = img.table.size() * sizeof(uint64_t); h.rows_bytes = img.blob_bytes;
    h.winspec_bytes = (img.windowed || img.has_pos) ? img.winspec.size() : 0; h.user_bytes = user_bytes;
    Section sec[4];
    sec[0].host = h.table_bytes ? (const void *)img.table.data() : (const void *)""; sec[0].bytes = h.table_bytes;
    sec[1].host = img.blob.data(); sec[1].bytes = h.rows_bytes;
    sec[2].host = h.winspec_bytes ? (const void *)img.winspec.data() : (const void *)""; sec[2].bytes = h.winspec_bytes;
    sec[3].host = user_bytes ? user : (const void *)""; sec[3].bytes = user_bytes;
    return write_image(path, h, sec, 0);
    RK_GUARD_END("rk_db_save_desc")
}

extern "C" int rk_db_image_info(const char *path, rk_db_info *info, uint64_t *user_bytes) {
    RK_GUARD_BEGIN
    MappedImage m;
    if (int rc = m.open_file(path, true, "rk_db_image_info")) return rc;
    if (info) info_from_header(m.h, info, -1);
    if (user_bytes) *user_bytes = m.h.user_bytes;
    return RK_OK;
    RK_GUARD_END("rk_db_image_info")
}

extern "C" int rk_db_image_user(const char *path, void *buf, uint64_t cap, uint64_t *len) {
    RK_GUARD_BEGIN
    if (!len) return fail(RK_ERR_INVALID, "rk_db_image_user: null argument");
    MappedImage m;
    if (int rc = m.open_file(path, false, "rk_db_image_user")) return rc;
    *len = m.h.user_bytes;
    if (buf && cap) memcpy(buf, m.base + m.off_user, (size_t)std::min<uint64_t>(cap, m.h.user_bytes));
    return RK_OK;
    RK_GUARD_END("rk_db_image_user")
}

extern "C" int rk_db_load(const char *path, int32_t device, rk_db **out) {
    RK_GUARD_BEGIN
    if (!out) return fail(RK_ERR_INVALID, "rk_db_load: null argument");
    *out = nullptr;
    MappedImage m;
    if (int rc = m.open_file(path, true, "rk_db_load")) return rc;  // size, header and payload checksums before the device is looked at
    const ImageHeader &h = m.h;
    int prev = 0;
    (void)hipGetDevice(&prev);
    struct Restore { int p; ~Restore() { (void)hipSetDevice(p); } } restore{prev};
    DbMeta meta{h.alphabet, h.convert_uo, h.k, h.n_branches, h.thr_log10, h.thr};
    rk_db *db = nullptr;
    if (int rc = open_db(meta, device, &db)) return rc;
#define LD_TRY(expr)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess) {                                                                   \
            int c_ = fail(e_ == hipErrorOutOfMemory ? RK_ERR_NOMEM : RK_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
            rk_db_destroy(db);                                                                    \
            return c_;                                                                            \
        }                                                                                         \
    } while (0)
    auto upload = [&](void *dst, uint64_t off, uint64_t bytes) -> hipError_t {  // in parts: the runtime stages pageable memory itself
        const uint64_t part = 1ull << 30;
        for (uint64_t o = 0; o < bytes; o += part) {
            const hipError_t e = hipMemcpy((unsigned char *)dst + o, m.base + off + o, (size_t)std::min(part, bytes - o), hipMemcpyHostToDevice);
            if (e != hipSuccess) return e;
        }
        return hipSuccess;
    };
    LD_TRY(hipMalloc(&db->d_table, h.table_bytes ? h.table_bytes : 8));
    LD_TRY(hipMalloc(&db->d_rows, h.rows_bytes));
    LD_TRY(upload(db->d_table, m.off_table, h.table_bytes));
    LD_TRY(upload(db->d_rows, m.off_rows, h.rows_bytes));
    if (h.windowed || h.has_pos) {
        LD_TRY(hipMalloc((void **)&db->d_winspec, h.winspec_bytes));
        LD_TRY(upload(db->d_winspec, m.off_winspec, h.winspec_bytes));
        db->windowed = h.windowed != 0;
        db->has_pos = h.has_pos != 0;
        if (h.windowed) {
            db->wp.W = h.wp_W; db->wp.n_win = h.wp_n_win; db->wp.s_stride = h.wp_s_stride; db->wp.main_cap = h.wp_main_cap; db->wp.work_cap = h.wp_work_cap;
            db->wp.stream = h.wp_stream != 0; db->wp.units_per_code = h.wp_units_per_code;
        }
    }
#undef LD_TRY
    db->compact_nib = h.compact_nib != 0;
    finish_db(db, meta, h.table_mode, h.indexed != 0, h.mono != 0, h.n_keys, h.n_entries, h.table_slots, h.hash_mask, h.table_bytes, h.rows_bytes, h.max_row_len);
    if (int rc = check_launchable(db)) { rk_db_destroy(db); return rc; }
    *out = db;
    return RK_OK;
    RK_GUARD_END("rk_db_load")
}
