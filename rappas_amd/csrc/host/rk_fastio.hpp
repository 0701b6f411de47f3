// rk_fastio.hpp -- the host side of `rk_place` at the engine's pace (round 4; SURVEY.md section 8(f) row N3: "at 10^8 reads/s the
// parser, not the kernel, becomes the bottleneck").  Header-only C++17, no placement compute.
//
// rk_hostio.hpp restates what the Java driver does around the native call one std::string at a time (3e6 reads/s parsed, 1.3e6
// deduplicated).  Here the same results come out of passes that every host thread works on at once:
//   * scan      the FASTA buffer is cut at record starts into one chunk per thread; a chunk's records become (header span, sequence
//               span, 128-bit hash of the gap-stripped sequence) without copying them -- a sequence written on one line is a span of the
//               file itself, multi-line records are joined in a side buffer of the chunk (src/inputs/FASTAPointer.java:137-149:
//               blank and '#' lines skipped, lines concatenated, the sequence trimmed);
//   * dedup     PlacementProcess.java:591-629 keeps the first read of every distinct gap-stripped sequence and hangs the later ones'
//               names on it.  Any 128-bit hash gives the same file as the reference's MD5 unless two different reads collide
//               (2^-128 per pair); records are bucketed by hash into shards, a shard is walked in file order by one thread (first
//               occurrence = first insertion).  --md5-dedup computes the reference's own digest instead (byte-compat tests);
//   * gather    the unique reads' characters, contiguous, for rk_place_batch;
//   * write     Main_PLACEMENT_v07.java:281-315 builds one JSON string and pushes it through seven regex replacements; the text
//               those produce is emitted directly, every thread formatting a range of reads (std::to_chars), the parts written at
//               their offsets of the output file.  Names that could interact with the reference's replacements (a ']' or '}' inside a
//               header) take the exact whole-document path of rk_hostio.hpp instead.
// tests/test_host_cpp.py and tests/test_gpu_hostio.py hold both paths to byte-identical files.
#pragma once
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include "rk_hostio.hpp"

namespace rkh {

// ------------------------------------------------------------------------------------------------------------------
// a few threads that live as long as the tool runs: run(fn) executes fn(t, T) on all of them and on the caller
// ------------------------------------------------------------------------------------------------------------------
class Team {
  public:
    explicit Team(unsigned threads) {
        if (threads < 1) threads = 1;
        for (unsigned i = 1; i < threads; i++) th_.emplace_back([this, i]() { loop(i); });
    }
    ~Team() {
        { std::lock_guard<std::mutex> lk(m_); stop_ = true; gen_++; }
        cv_.notify_all();
        for (std::thread &t : th_) t.join();
    }
    unsigned size() const { return (unsigned)th_.size() + 1; }
    void run(const std::function<void(unsigned, unsigned)> &fn) {
        if (th_.empty()) { fn(0, 1); return; }
        {
            std::lock_guard<std::mutex> lk(m_);
            fn_ = &fn; left_ = (unsigned)th_.size(); gen_++;
        }
        cv_.notify_all();
        std::exception_ptr mine;
        try { fn(0, size()); } catch (...) { mine = std::current_exception(); }
        {
            std::unique_lock<std::mutex> lk(m_);
            done_.wait(lk, [&]() { return left_ == 0; });
        }
        if (mine) std::rethrow_exception(mine);
        if (err_) { std::exception_ptr e = err_; err_ = nullptr; std::rethrow_exception(e); }
    }

  private:
    void loop(unsigned me) {
        uint64_t seen = 0;
        while (true) {
            const std::function<void(unsigned, unsigned)> *fn;
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&]() { return gen_ != seen; });
                seen = gen_;
                if (stop_) return;
                fn = fn_;
            }
            try { (*fn)(me, size()); } catch (...) { std::lock_guard<std::mutex> lk(m_); if (!err_) err_ = std::current_exception(); }
            {
                std::lock_guard<std::mutex> lk(m_);
                if (--left_ == 0) done_.notify_all();
            }
        }
    }
    std::vector<std::thread> th_;
    std::mutex m_;
    std::condition_variable cv_, done_;
    const std::function<void(unsigned, unsigned)> *fn_ = nullptr;
    std::exception_ptr err_;
    uint64_t gen_ = 0;
    unsigned left_ = 0;
    bool stop_ = false;
};

// ------------------------------------------------------------------------------------------------------------------
// a file mapped read-only (the FASTA; page cache -> address space, nothing is copied)
// ------------------------------------------------------------------------------------------------------------------
struct MappedFile {
    const char *data = nullptr;
    size_t size = 0;
    MappedFile() = default;
    MappedFile(const MappedFile &) = delete;
    MappedFile &operator=(const MappedFile &) = delete;
    ~MappedFile() { if (data && size) (void)munmap((void *)data, size); }
    void open_file(const std::string &path) {
        const int fd = open(path.c_str(), O_RDONLY | O_CLOEXEC);
        if (fd < 0) throw std::runtime_error("cannot open " + path);
        struct stat st;
        if (fstat(fd, &st) != 0) { (void)close(fd); throw std::runtime_error("cannot stat " + path); }
        size = (size_t)st.st_size;
        if (size) {
            void *m = mmap(nullptr, size, PROT_READ, MAP_PRIVATE | MAP_POPULATE, fd, 0);
            if (m == MAP_FAILED) { (void)close(fd); size = 0; throw std::runtime_error("cannot map " + path); }
            data = (const char *)m;
        }
        (void)close(fd);
    }
};

// ------------------------------------------------------------------------------------------------------------------
// large scratch arrays: anonymous mappings with huge pages asked for -- a gigabyte first touched by 32 threads of one process through
// 4 KB page faults spends more time in the kernel's fault path than the passes spend on the data
// ------------------------------------------------------------------------------------------------------------------
struct BigBuf {
    void *p = nullptr;
    size_t bytes = 0;
    BigBuf() = default;
    BigBuf(const BigBuf &) = delete;
    BigBuf &operator=(const BigBuf &) = delete;
    BigBuf(BigBuf &&o) noexcept : p(o.p), bytes(o.bytes) { o.p = nullptr; o.bytes = 0; }
    BigBuf &operator=(BigBuf &&o) noexcept { release(); p = o.p; bytes = o.bytes; o.p = nullptr; o.bytes = 0; return *this; }
    ~BigBuf() { release(); }
    void release() { if (p) (void)munmap(p, bytes); p = nullptr; bytes = 0; }
    void alloc(size_t n) {
        release();
        const size_t huge = (size_t)2 << 20;
        bytes = ((n ? n : 1) + huge - 1) & ~(huge - 1);
        void *m = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (m == MAP_FAILED) { bytes = 0; throw std::bad_alloc(); }
        p = m;
#ifdef MADV_HUGEPAGE
        (void)madvise(p, bytes, MADV_HUGEPAGE);
#endif
    }
};
template <class T>
struct RawArray {  // n elements of a trivially copyable T, not filled
    BigBuf buf;
    size_t n = 0;
    void alloc(size_t m) { buf.alloc(m * sizeof(T)); n = m; }
    T *data() { return (T *)buf.p; }
    const T *data() const { return (const T *)buf.p; }
    T &operator[](size_t i) { return ((T *)buf.p)[i]; }
    const T &operator[](size_t i) const { return ((const T *)buf.p)[i]; }
    size_t size() const { return n; }
};

// ------------------------------------------------------------------------------------------------------------------
// scan
// ------------------------------------------------------------------------------------------------------------------
struct Hash128 {
    uint64_t lo = 0, hi = 0;
    bool operator==(const Hash128 &o) const { return lo == o.lo && hi == o.hi; }
};

// Two independent 64-bit multiply-fold lanes over 8-byte words (the last, partial word zero-padded, the length mixed in): a dedup
// key, not a cryptographic digest.
inline uint64_t fold64(uint64_t a, uint64_t b) {
    const unsigned __int128 p = (unsigned __int128)a * b;
    return (uint64_t)p ^ (uint64_t)(p >> 64);
}
inline Hash128 hash_bytes(const char *p, size_t n) {
    uint64_t a = 0x9E3779B97F4A7C15ull ^ n, b = 0xD6E8FEB86659FD93ull + n;
    size_t i = 0;
    for (; i + 8 <= n; i += 8) {
        uint64_t w;
        memcpy(&w, p + i, 8);
        a = fold64(a ^ w, 0xA0761D6478BD642Full);
        b = fold64(b ^ w, 0xE7037ED1A0B428DBull) + 0x8EBC6AF09C88C6E3ull;
    }
    uint64_t w = 0;
    if (i < n) memcpy(&w, p + i, n - i);
    a = fold64(a ^ w, 0x589965CC75374CC3ull);
    b = fold64(b ^ w, 0x1D8E4E27C47D124Full);
    return Hash128{fold64(a, 0x2D358DCCAA6C78A5ull) ^ b, fold64(b, 0x8BB84B93962EACC9ull) ^ a};
}

struct Record {
    const char *hdr;  // header line behind '>', without the line's '\r'
    const char *seq;  // trimmed sequence: a span of the file, or of the chunk's side buffer (multi-line records)
    uint32_t hdr_len, seq_len;
};

// joined multi-line sequences of a chunk: blocks that never move (the records point into them)
struct SideBuf {
    std::vector<std::unique_ptr<char[]>> blocks;
    size_t used = 0, cap = 0;
    char *cur = nullptr;
    // room for n more bytes behind the `keep` bytes of the record being joined (which move along if a new block is needed)
    char *grow(size_t keep, size_t n) {
        if (used + n > cap) {
            const size_t want = std::max<size_t>(1u << 20, 2 * (keep + n));
            std::unique_ptr<char[]> blk(new char[want]);
            if (keep) memcpy(blk.get(), cur + used - keep, keep);
            blocks.push_back(std::move(blk));
            cur = blocks.back().get();
            cap = want;
            used = keep;
        }
        return cur + used;
    }
};

struct FastaScan {
    RawArray<Record> recs;               // file order
    RawArray<Hash128> hash;              // of the gap-stripped sequence (or its MD5)
    std::vector<SideBuf> side;           // per chunk: joined multi-line sequences (the records point into these)
    uint64_t bytes = 0;
};

// one chunk [lo, hi) of the text that starts at a record ('>' at a line start) or at the file's start
inline size_t scan_chunk(const char *text, size_t lo, size_t hi, bool md5_digest, Record *recs, Hash128 *hash, SideBuf &side) {
    size_t n_out = 0;
    std::string nogap;
    auto close_record = [&](Record &r, const char *s, size_t n) {
        while (n && (unsigned char)s[0] <= ' ') { s++; n--; }          // String.trim()
        while (n && (unsigned char)s[n - 1] <= ' ') n--;
        r.seq = s;
        r.seq_len = (uint32_t)n;
        const char *hs = s;
        size_t hn = n;
        if (memchr(s, '-', n)) {  // sequence.replaceAll("-", "") (PlacementProcess.java:592)
            nogap.clear();
            for (size_t i = 0; i < n; i++)
                if (s[i] != '-') nogap.push_back(s[i]);
            hs = nogap.data();
            hn = nogap.size();
        }
        Hash128 h;
        if (md5_digest) {
            const std::array<uint8_t, 16> d = md5(std::string(hs, hn));
            memcpy(&h.lo, d.data(), 8);
            memcpy(&h.hi, d.data() + 8, 8);
        } else {
            h = hash_bytes(hs, hn);
        }
        recs[n_out] = r;
        hash[n_out] = h;
        n_out++;
    };
    size_t pos = lo;
    bool open = false;
    Record cur{};
    const char *first_line = nullptr;
    size_t first_len = 0, lines = 0, joined = 0;  // joined: bytes of the current record in the side buffer
    auto finish = [&]() {
        if (lines <= 1) close_record(cur, first_line, first_len);
        else close_record(cur, side.cur + side.used - joined, joined);
    };
    while (pos < hi) {
        const char *nl = (const char *)memchr(text + pos, '\n', hi - pos);
        const size_t end = nl ? (size_t)(nl - text) : hi;
        size_t e = end;
        if (e > pos && text[e - 1] == '\r') e--;
        if (e > pos && text[pos] != '#') {
            if (text[pos] == '>') {
                if (open) finish();
                cur.hdr = text + pos + 1;
                cur.hdr_len = (uint32_t)(e - pos - 1);
                open = true;
                lines = 0; first_line = text + pos; first_len = 0; joined = 0;
            } else if (open) {
                if (lines == 0) { first_line = text + pos; first_len = e - pos; }
                else {
                    if (lines == 1) { memcpy(side.grow(0, first_len), first_line, first_len); side.used += first_len; joined = first_len; }
                    memcpy(side.grow(joined, e - pos), text + pos, e - pos);
                    side.used += e - pos;
                    joined += e - pos;
                }
                lines++;
            }
        }
        pos = end + 1;
    }
    if (open) finish();
    return n_out;
}

// records a chunk will yield: lines that open with '>' (the same test scan_chunk applies)
inline size_t count_records(const char *text, size_t lo, size_t hi) {
    size_t n = 0, pos = lo;
    while (pos < hi) {
        if (text[pos] == '>') n++;
        const char *nl = (const char *)memchr(text + pos, '\n', hi - pos);
        if (!nl) break;
        pos = (size_t)(nl - text) + 1;
    }
    return n;
}

inline FastaScan scan_fasta(const char *text, size_t size, Team &team, bool md5_digest = false) {
    const unsigned T = team.size();
    // chunk starts: the first record start at or behind the nominal cut (a '>' that opens a line)
    std::vector<size_t> cut(T + 1, size);
    cut[0] = 0;
    for (unsigned t = 1; t < T; t++) {
        size_t p = size / T * t;
        if (p < cut[t - 1]) p = cut[t - 1];
        while (p < size) {
            if (text[p] == '>' && (p == 0 || text[p - 1] == '\n')) break;
            const char *nl = (const char *)memchr(text + p, '\n', size - p);
            if (!nl) { p = size; break; }
            p = (size_t)(nl - text) + 1;
        }
        cut[t] = p;
    }
    FastaScan out;
    out.side.resize(T);
    out.bytes = size;
    // records per chunk first (one memchr pass), so that every thread writes its records where they belong: no growing vectors,
    // no concatenation
    std::vector<size_t> base(T + 1, 0);
    team.run([&](unsigned t, unsigned) { base[t + 1] = count_records(text, cut[t], cut[t + 1]); });
    for (unsigned t = 0; t < T; t++) base[t + 1] += base[t];
    out.recs.alloc(base[T]);
    out.hash.alloc(base[T]);
    team.run([&](unsigned t, unsigned) {
        const size_t lo = cut[t], hi = cut[t + 1];
        if (hi <= lo) return;
        const size_t got = scan_chunk(text, lo, hi, md5_digest, out.recs.data() + base[t], out.hash.data() + base[t], out.side[t]);
        if (got != base[t + 1] - base[t]) throw std::runtime_error("internal: FASTA record count changed between the two passes");
    });
    return out;
}

// ------------------------------------------------------------------------------------------------------------------
// dedup: first_of[i] = the first record with record i's sequence; unique reads numbered in file order of their first record; the
// later records of a sequence chained in file order (next_dup)
// ------------------------------------------------------------------------------------------------------------------
struct FastDedup {
    RawArray<uint32_t> uniq_of_rec;  // [n records]
    RawArray<uint32_t> first_rec;    // [n unique]
    RawArray<uint32_t> next_dup;     // [n records] the next record with the same sequence (file order), 0xFFFFFFFF = none
    uint64_t n_dups = 0;
};

inline FastDedup dedup_fast(const FastaScan &sc, Team &team) {
    const size_t n = sc.recs.size();
    if (n >= 0xFFFFFFFFull) throw std::runtime_error("more than 2^32 - 2 reads in one file");
    const unsigned T = team.size();
    constexpr unsigned LOGS = 8, S = 1u << LOGS;  // shards by the hash's top bits
    FastDedup d;
    d.uniq_of_rec.alloc(n);
    d.next_dup.alloc(n);
    RawArray<uint32_t> first_of, order;
    first_of.alloc(n);
    order.alloc(n);
    // records per (thread, shard), then every shard's index list: thread-major inside a shard = file order
    std::vector<std::vector<uint32_t>> cnt(T, std::vector<uint32_t>(S, 0));
    team.run([&](unsigned t, unsigned) {
        const size_t lo = n * t / T, hi = n * (t + 1) / T;
        for (size_t i = lo; i < hi; i++) {
            cnt[t][sc.hash[i].hi >> (64 - LOGS)]++;
            d.uniq_of_rec[i] = 0;
            d.next_dup[i] = 0xFFFFFFFFu;
        }
    });
    std::vector<size_t> shard_base(S + 1, 0);
    std::vector<std::vector<size_t>> at(T, std::vector<size_t>(S, 0));
    {
        size_t run = 0;
        for (unsigned s = 0; s < S; s++) {
            shard_base[s] = run;
            for (unsigned t = 0; t < T; t++) { at[t][s] = run; run += cnt[t][s]; }
        }
        shard_base[S] = run;
    }
    team.run([&](unsigned t, unsigned) {
        const size_t lo = n * t / T, hi = n * (t + 1) / T;
        std::vector<size_t> cur = at[t];
        for (size_t i = lo; i < hi; i++) order[cur[sc.hash[i].hi >> (64 - LOGS)]++] = (uint32_t)i;
    });
    std::atomic<unsigned> next_shard{0};
    std::atomic<uint64_t> dups{0};
    team.run([&](unsigned, unsigned) {
        std::vector<uint32_t> table, tail;
        uint64_t my_dups = 0;
        while (true) {
            const unsigned s = next_shard.fetch_add(1);
            if (s >= S) break;
            const size_t lo = shard_base[s], m = shard_base[s + 1] - lo;
            if (!m) continue;
            size_t cap = 16;
            while (cap < 2 * m) cap <<= 1;
            table.assign(cap, 0xFFFFFFFFu);  // open addressing: the first record of a sequence
            for (size_t j = 0; j < m; j++) {
                const uint32_t i = order[lo + j];
                const Hash128 h = sc.hash[i];
                size_t slot = (size_t)h.lo & (cap - 1);
                while (true) {
                    const uint32_t f = table[slot];
                    if (f == 0xFFFFFFFFu) { table[slot] = i; first_of[i] = i; break; }
                    if (sc.hash[f] == h) {
                        first_of[i] = f;
                        // chain: the tail of f's list is kept in next_dup[f]'s own chain end -- walk-free through a tail slot per first
                        // record, stored in uniq_of_rec[f] until the ids are assigned
                        const uint32_t last = d.uniq_of_rec[f] ? d.uniq_of_rec[f] - 1u : f;
                        d.next_dup[last] = i;
                        d.uniq_of_rec[f] = i + 1u;
                        my_dups++;
                        break;
                    }
                    slot = (slot + 1) & (cap - 1);
                }
            }
        }
        dups.fetch_add(my_dups);
    });
    d.n_dups = dups.load();
    // ids in file order of the first records
    std::vector<size_t> firsts(T + 1, 0);
    team.run([&](unsigned t, unsigned) {
        const size_t lo = n * t / T, hi = n * (t + 1) / T;
        size_t c = 0;
        for (size_t i = lo; i < hi; i++) c += first_of[i] == i;
        firsts[t + 1] = c;
    });
    for (unsigned t = 0; t < T; t++) firsts[t + 1] += firsts[t];
    d.first_rec.alloc(firsts[T]);
    team.run([&](unsigned t, unsigned) {
        const size_t lo = n * t / T, hi = n * (t + 1) / T;
        size_t u = firsts[t];
        for (size_t i = lo; i < hi; i++)
            if (first_of[i] == i) { d.first_rec[u] = (uint32_t)i; d.uniq_of_rec[i] = (uint32_t)u; u++; }
    });
    team.run([&](unsigned t, unsigned) {
        const size_t lo = n * t / T, hi = n * (t + 1) / T;
        for (size_t i = lo; i < hi; i++)
            if (first_of[i] != i) d.uniq_of_rec[i] = d.uniq_of_rec[first_of[i]];
    });
    return d;
}

// the unique reads' characters, contiguous (what rk_place_batch takes).  Buffers are allocated without being filled (a std::vector
// would write zeros over ~600 MB on one thread first); every thread touches its own part first.
inline void gather_unique(const FastaScan &sc, const FastDedup &d, Team &team, RawArray<char> &seq, RawArray<uint64_t> &off) {
    const size_t n = d.first_rec.size();
    const unsigned T = team.size();
    off.alloc(n + 1);
    std::vector<uint64_t> base(T + 1, 0);
    team.run([&](unsigned t, unsigned) {
        const size_t lo = n * t / T, hi = n * (t + 1) / T;
        uint64_t sum = 0;
        for (size_t i = lo; i < hi; i++) sum += sc.recs[d.first_rec[i]].seq_len;
        base[t + 1] = sum;
    });
    for (unsigned t = 0; t < T; t++) base[t + 1] += base[t];
    seq.alloc(base[T]);
    team.run([&](unsigned t, unsigned) {
        const size_t lo = n * t / T, hi = n * (t + 1) / T;
        uint64_t at = base[t];
        for (size_t i = lo; i < hi; i++) {
            const Record &r = sc.recs[d.first_rec[i]];
            off.data()[i] = at;
            memcpy(seq.data() + at, r.seq, r.seq_len);
            at += r.seq_len;
        }
        if (t + 1 == T) off.data()[n] = base[T];
    });
}

// ------------------------------------------------------------------------------------------------------------------
// write: the text Main_PLACEMENT_v07.java:281-315 ends up with, emitted directly
// ------------------------------------------------------------------------------------------------------------------
// Can the reference's replacements ("},{"  "],\""  "]}],"  "],["  ...) touch this string once it is a JSON literal?  Only through a
// ']' or a '}' (every pattern holds one); such names take the exact whole-document path.
inline bool plain_for_prettifier(const char *s, size_t n) { return !memchr(s, ']', n) && !memchr(s, '}', n); }

inline void append_jstr(std::string &o, const char *s, size_t n) {  // rkh::jstr without the temporary
    bool simple = true;
    for (size_t i = 0; i < n; i++) {
        const unsigned char c = (unsigned char)s[i];
        if (c < 0x20 || c == '"' || c == '\\' || c == '/' || c >= 0x7F) { simple = false; break; }
    }
    if (simple) { o.push_back('"'); o.append(s, n); o.push_back('"'); }
    else o += jstr(std::string(s, n));
}

// java_layout (rk_hostio.hpp) without a temporary: `sci` is std::to_chars' scientific form ([-]d[.ddd]e[+-]XX, `len` characters), the
// result -- Float.toString / Double.toString's layout of the same shortest digits -- is written at `o`; returns its end (<= 40 bytes)
inline char *java_layout_to(char *o, const char *sci, size_t len, double value) {
    if (value == 0) {
        if (std::signbit(value)) *o++ = '-';
        memcpy(o, "0.0", 3);
        return o + 3;
    }
    const char *p = sci, *end = sci + len;
    if (*p == '-') { *o++ = '-'; p++; }
    char ds[40];
    int nd = 0;
    while (p < end && *p != 'e') { if (*p != '.') ds[nd++] = *p; p++; }
    int exp10 = 0;
    {
        p++;
        bool en = false;
        if (p < end && *p == '-') { en = true; p++; } else if (p < end && *p == '+') p++;
        while (p < end) exp10 = exp10 * 10 + (*p++ - '0');
        if (en) exp10 = -exp10;
    }
    while (nd > 1 && ds[nd - 1] == '0') nd--;
    const int e10 = exp10 + 1;  // value = 0.ds * 10^e10
    const double a = std::fabs(value);
    if (a >= 1e-3 && a < 1e7) {
        if (e10 <= 0) {
            *o++ = '0'; *o++ = '.';
            for (int i = 0; i < -e10; i++) *o++ = '0';
            memcpy(o, ds, (size_t)nd); o += nd;
        } else if (e10 >= nd) {
            memcpy(o, ds, (size_t)nd); o += nd;
            for (int i = 0; i < e10 - nd; i++) *o++ = '0';
            *o++ = '.'; *o++ = '0';
        } else {
            memcpy(o, ds, (size_t)e10); o += e10;
            *o++ = '.';
            memcpy(o, ds + e10, (size_t)(nd - e10)); o += nd - e10;
        }
    } else {
        *o++ = ds[0]; *o++ = '.';
        if (nd > 1) { memcpy(o, ds + 1, (size_t)(nd - 1)); o += nd - 1; } else *o++ = '0';
        *o++ = 'E';
        o = std::to_chars(o, o + 8, e10 - 1).ptr;
    }
    return o;
}
template <class F>
inline void append_java_number(std::string &o, F v) {  // java_float_to_string / java_double_to_string appended to o
    if (!std::isfinite(v)) { o += "null"; return; }
    char sci[48], out[56];
    const auto r = std::to_chars(sci, sci + sizeof(sci), v, std::chars_format::scientific);
    o.append(out, (size_t)(java_layout_to(out, sci, (size_t)(r.ptr - sci), (double)v) - out));
}

struct FastWriteStats {
    uint64_t placed = 0, bytes = 0;
    bool exact_path = false;  // some name needed the whole-document replacements
    double format_s = 0, io_s = 0;
};

// jplace_placements + jplace_document of rk_hostio.hpp in one go, written to `path`.  names of unique read u: the first record's
// full header, then -- PlacementProcess.java:594-598,618 -- the later records' headers cut at the first space, in file order.
inline FastWriteStats write_jplace_fast(const std::string &path, const Tree &t, const FastaScan &sc, const FastDedup &d, uint32_t K,
                                        const uint8_t *n_rows, const uint16_t *branch, const float *score, const double *lwr,
                                        const std::string &call_string, bool guppy, Team &team) {
    const size_t n = d.first_rec.size();
    // the skeleton around the placements: the exact function on an empty and on a one-placement document gives head, opening and tail
    const std::string empty_doc = jplace_document(t, {}, call_string, guppy);
    Placement probe;
    probe.rows.push_back({"@1", "@2", "@3", "@4", "@5"});
    probe.names.push_back("@n");
    const std::string one_doc = jplace_document(t, {probe}, call_string, guppy);
    const std::string marker = "\"p\":\n\t[[@1,@2,@3,@4,@5]],\n\t\"nm\":\n\t[[\"@n\",1]]";
    const size_t at = one_doc.find(marker);
    if (at == std::string::npos) throw std::runtime_error("internal: jplace skeleton not recognised");
    const std::string head = one_doc.substr(0, at), tail = one_doc.substr(at + marker.size());
    // per-node strings: edge number, distal length
    std::vector<std::string> edge(t.nodes.size()), distal(t.nodes.size());
    for (size_t b = 0; b < t.nodes.size(); b++) {
        edge[b] = std::to_string(t.nodes[b].jplace_edge);
        distal[b] = java_float_to_string(t.nodes[b].bl / 2.0f);
    }
    const unsigned T = team.size();
    // a thread's text goes into a mapping of its own, sized for the worst case (untouched pages cost nothing): no growing strings
    struct Part { BigBuf buf; size_t len = 0; bool empty() const { return len == 0; } const char *data() const { return (const char *)buf.p; } size_t size() const { return len; } };
    std::vector<Part> part(T);
    std::vector<uint64_t> placed(T, 0);
    std::atomic<bool> needs_exact{false};
    auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_f0 = now();
    size_t max_edge = 1, max_distal = 3;
    for (size_t b = 0; b < t.nodes.size(); b++) { max_edge = std::max(max_edge, edge[b].size()); max_distal = std::max(max_distal, distal[b].size()); }
    team.run([&](unsigned tt, unsigned) {
        const size_t lo = n * tt / T, hi = n * (tt + 1) / T;
        size_t bound = 64;
        for (size_t u = lo; u < hi; u++) {
            if (!n_rows[u]) continue;
            bound += 48 + (size_t)n_rows[u] * (max_edge + max_distal + 15 + 25 + 12);
            for (uint32_t rec = d.first_rec[u]; rec != 0xFFFFFFFFu; rec = d.next_dup[rec]) bound += 6 * (size_t)sc.recs[rec].hdr_len + 12;  // (an escaped character: <= 6)
        }
        part[tt].buf.alloc(bound);
        char *const o0 = (char *)part[tt].buf.p;
        char *o = o0;
        auto lit = [&](const char *str, size_t len) { memcpy(o, str, len); o += len; };
#define LIT(str_) lit(str_, sizeof(str_) - 1)
        auto num = [&](auto v) {
            if (!std::isfinite(v)) { LIT("null"); return; }
            char sci[48];
            const auto r = std::to_chars(sci, sci + sizeof(sci), v, std::chars_format::scientific);
            o = java_layout_to(o, sci, (size_t)(r.ptr - sci), (double)v);
        };
        std::string esc;
        for (size_t u = lo; u < hi; u++) {
            if (!n_rows[u]) continue;
            // (every placement but the document's first is preceded by the separator the replacements make of "},{")
            LIT("\n},{\n\t\"p\":\n\t[");
            for (uint32_t j = 0; j < n_rows[u]; j++) {
                const uint32_t b = branch[u * K + j];
                if (b >= t.nodes.size()) throw std::runtime_error("placement on branch " + std::to_string(b) + " which the tree does not have");
                if (j) LIT(",\n\t");
                *o++ = '[';
                const float sv = score[u * K + j];
                const double lv = lwr[u * K + j];
                if (guppy) { lit(distal[b].data(), distal[b].size()); *o++ = ','; lit(edge[b].data(), edge[b].size()); *o++ = ','; num(lv); *o++ = ','; num(sv); }
                else { lit(edge[b].data(), edge[b].size()); *o++ = ','; num(sv); *o++ = ','; num(lv); *o++ = ','; lit(distal[b].data(), distal[b].size()); }
                LIT(",0.0]");
            }
            LIT("],\n\t\"nm\":\n\t[");
            uint32_t rec = d.first_rec[u];
            bool first = true;
            while (rec != 0xFFFFFFFFu) {
                const Record &r = sc.recs[rec];
                size_t len = r.hdr_len;
                if (!first) {
                    const void *sp = memchr(r.hdr, ' ', r.hdr_len);
                    if (sp) len = (size_t)((const char *)sp - r.hdr);
                    LIT(",\n\t");
                }
                if (!plain_for_prettifier(r.hdr, len)) needs_exact.store(true, std::memory_order_relaxed);
                *o++ = '[';
                esc.clear();
                append_jstr(esc, r.hdr, len);
                lit(esc.data(), esc.size());
                LIT(",1]");
                first = false;
                rec = d.next_dup[rec];
            }
            *o++ = ']';
            placed[tt]++;
        }
#undef LIT
        part[tt].len = (size_t)(o - o0);
        if (part[tt].len > bound) throw std::runtime_error("internal: jplace part outgrew its bound");
    });
    FastWriteStats st;
    st.format_s = now() - t_f0;
    const double t_w0 = now();
    for (unsigned tt = 0; tt < T; tt++) st.placed += placed[tt];
    if (needs_exact.load()) {
        // the exact path of rk_hostio.hpp (a header with ']' or '}': the reference's replacements may reach into it)
        std::vector<Placement> pl;
        for (size_t u = 0; u < n; u++) {
            if (!n_rows[u]) continue;
            Placement p;
            for (uint32_t j = 0; j < n_rows[u]; j++) {
                const uint32_t b = branch[u * K + j];
                const std::string like = java_float_to_string(score[u * K + j]), ratio = java_double_to_string(lwr[u * K + j]);
                if (guppy) p.rows.push_back({distal[b], edge[b], ratio, like, "0.0"});
                else p.rows.push_back({edge[b], like, ratio, distal[b], "0.0"});
            }
            uint32_t rec = d.first_rec[u];
            bool first = true;
            while (rec != 0xFFFFFFFFu) {
                const Record &r = sc.recs[rec];
                std::string name(r.hdr, r.hdr_len);
                if (!first) { const size_t cut = name.find(' '); if (cut != std::string::npos) name.resize(cut); }
                p.names.push_back(std::move(name));
                first = false;
                rec = d.next_dup[rec];
            }
            pl.push_back(std::move(p));
        }
        const std::string doc = jplace_document(t, pl, call_string, guppy);
        FILE *f = fopen(path.c_str(), "wb");
        if (!f || fwrite(doc.data(), 1, doc.size(), f) != doc.size() || fclose(f) != 0) throw std::runtime_error("cannot write " + path);
        st.bytes = doc.size();
        st.exact_path = true;
        return st;
    }
    if (st.placed == 0) {
        FILE *f = fopen(path.c_str(), "wb");
        if (!f || fwrite(empty_doc.data(), 1, empty_doc.size(), f) != empty_doc.size() || fclose(f) != 0) throw std::runtime_error("cannot write " + path);
        st.bytes = empty_doc.size();
        return st;
    }
    // the document's first placement has no separator in front of it: `head` ends where its "p" starts
    static const std::string sep = "\n},{\n\t";
    unsigned first_part = 0;
    while (part[first_part].empty()) first_part++;
    std::vector<uint64_t> at_off(T + 1, 0);
    at_off[0] = head.size();
    for (unsigned tt = 0; tt < T; tt++) at_off[tt + 1] = at_off[tt] + part[tt].size() - (tt == first_part ? sep.size() : 0);
    const uint64_t total = at_off[T] + tail.size();
    // the output file is mapped and every thread copies its part in: page-cache pages are then allocated by all threads at once
    // (write() / pwrite() on one file take the inode's lock in turn: 3 - 4 GB/s whatever the number of threads)
    const int fd = open(path.c_str(), O_RDWR | O_CREAT | O_TRUNC | O_CLOEXEC, 0644);
    if (fd < 0) throw std::runtime_error("cannot write " + path);
    char *map = nullptr;
    try {
        if (ftruncate(fd, (off_t)total) != 0) throw std::runtime_error("cannot size " + path);
        void *m = mmap(nullptr, (size_t)total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        if (m == MAP_FAILED) throw std::runtime_error("cannot map " + path);
        map = (char *)m;
        team.run([&](unsigned tt, unsigned) {
            if (tt == 0) { memcpy(map, head.data(), head.size()); memcpy(map + at_off[T], tail.data(), tail.size()); }
            if (part[tt].empty()) return;
            const size_t skip = tt == first_part ? sep.size() : 0;
            char *dst = map + at_off[tt];
            const size_t len = part[tt].size() - skip;
#ifdef MADV_POPULATE_WRITE
            {   // the range's pages in one call instead of one fault each (Linux >= 5.14; ignored elsewhere)
                const uintptr_t a0 = ((uintptr_t)dst + 4095) & ~(uintptr_t)4095, a1 = ((uintptr_t)dst + len) & ~(uintptr_t)4095;
                if (a1 > a0) (void)madvise((void *)a0, a1 - a0, MADV_POPULATE_WRITE);
            }
#endif
            memcpy(dst, part[tt].data() + skip, len);
        });
        if (munmap(map, (size_t)total) != 0) { map = nullptr; throw std::runtime_error("cannot unmap " + path); }
        map = nullptr;
    } catch (...) {
        if (map) (void)munmap(map, (size_t)total);
        (void)close(fd);
        throw;
    }
    if (close(fd) != 0) throw std::runtime_error("cannot close " + path);
    st.bytes = total;
    st.io_s = now() - t_w0;
    return st;
}

// notplaced_<query>.tsv of rk_hostio.hpp: notplaced_log over the scan's records
inline std::string notplaced_log_fast(const FastaScan &sc, const FastDedup &d, const uint32_t *flags) {
    std::string out;
    for (size_t i = 0; i < sc.recs.size(); i++)
        if (!(flags[d.uniq_of_rec[i]] & 1u)) { out.append(sc.recs[i].hdr, sc.recs[i].hdr_len); out.push_back('\n'); }
    return out;
}

// ------------------------------------------------------------------------------------------------------------------
// the reference tree as the `user` blob of a database image (rk_db_save): exact, line based
//   RKTREE 1 <nodes> <root>
//   <id> <parent> <jplace edge> <branch length, float32 bits in hex> <n children> <child ids...> \t <label>
// ------------------------------------------------------------------------------------------------------------------
inline std::string tree_to_blob(const Tree &t) {
    std::string o = "RKTREE 1 " + std::to_string(t.nodes.size()) + " " + std::to_string(t.root) + "\n";
    char buf[32];
    for (const Node &n : t.nodes) {
        uint32_t bits;
        memcpy(&bits, &n.bl, 4);
        snprintf(buf, sizeof(buf), "%08x", bits);
        o += std::to_string(n.id) + " " + std::to_string(n.parent) + " " + std::to_string(n.jplace_edge) + " " + buf + " " + std::to_string(n.children.size());
        for (int c : n.children) o += " " + std::to_string(c);
        o += "\t" + n.label + "\n";
    }
    return o;
}
inline Tree tree_from_blob(const std::string &blob) {
    Tree t;
    size_t pos = blob.find('\n');
    if (pos == std::string::npos || blob.compare(0, 9, "RKTREE 1 ") != 0) throw std::runtime_error("database image carries no reference tree (RKTREE blob)");
    size_t n_nodes = 0;
    int root = 0;
    if (sscanf(blob.c_str() + 9, "%zu %d", &n_nodes, &root) != 2 || n_nodes == 0 || n_nodes > 65535 || root < 0 || (size_t)root >= n_nodes)
        throw std::runtime_error("database image: malformed tree header");
    t.nodes.resize(n_nodes);
    t.root = root;
    pos++;
    for (size_t i = 0; i < n_nodes; i++) {
        const size_t nl = blob.find('\n', pos);
        if (nl == std::string::npos) throw std::runtime_error("database image: tree cut short");
        const size_t tab = blob.find('\t', pos);
        if (tab == std::string::npos || tab > nl) throw std::runtime_error("database image: malformed tree line");
        Node &n = t.nodes[i];
        const std::string nums = blob.substr(pos, tab - pos);
        char *e = nullptr;
        const char *p = nums.c_str();
        n.id = (int)strtol(p, &e, 10); p = e;
        n.parent = (int)strtol(p, &e, 10); p = e;
        n.jplace_edge = (int)strtol(p, &e, 10); p = e;
        const uint32_t bits = (uint32_t)strtoul(p, &e, 16); p = e;
        memcpy(&n.bl, &bits, 4);
        const long nc = strtol(p, &e, 10); p = e;
        if (n.id != (int)i || nc < 0 || nc > 65535 || n.parent < -1 || n.parent >= (int)n_nodes) throw std::runtime_error("database image: malformed tree line");
        for (long c = 0; c < nc; c++) {
            const long ch = strtol(p, &e, 10);
            if (e == p || ch < 0 || (size_t)ch >= n_nodes) throw std::runtime_error("database image: malformed tree line");
            p = e;
            n.children.push_back((int)ch);
        }
        n.label = blob.substr(tab + 1, nl - tab - 1);
        pos = nl + 1;
    }
    return t;
}

}  // namespace rkh
