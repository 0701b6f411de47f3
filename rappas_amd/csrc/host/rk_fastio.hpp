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
#include <condition_variable>
#include <functional>
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

struct FastaScan {
    std::vector<Record> recs;            // file order
    std::vector<Hash128> hash;           // of the gap-stripped sequence (or its MD5)
    std::vector<std::string> side;       // per chunk: joined multi-line sequences (the records point into these)
    uint64_t bytes = 0;
};

// one chunk [lo, hi) of the text that starts at a record ('>' at a line start) or at the file's start
inline void scan_chunk(const char *text, size_t lo, size_t hi, bool md5_digest, std::vector<Record> &recs, std::vector<Hash128> &hash,
                       std::string &side) {
    // first pass over the chunk: how much side buffer the multi-line records need (so that it never reallocates: records point into it)
    {
        size_t need = 0;
        size_t pos = lo;
        bool open = false;
        size_t lines = 0, len = 0;
        while (pos < hi) {
            const char *nl = (const char *)memchr(text + pos, '\n', hi - pos);
            const size_t end = nl ? (size_t)(nl - text) : hi;
            size_t e = end;
            if (e > pos && text[e - 1] == '\r') e--;
            if (e > pos && text[pos] != '#') {
                if (text[pos] == '>') { if (open && lines > 1) need += len; open = true; lines = 0; len = 0; }
                else if (open) { lines++; len += e - pos; }
            }
            pos = end + 1;
        }
        if (open && lines > 1) need += len;
        side.clear();
        side.reserve(need + 1);
    }
    std::string nogap;
    auto close_record = [&](Record &r, const char *first_line, size_t first_len, size_t lines, size_t side_start) {
        const char *s;
        size_t n;
        if (lines <= 1) { s = first_line; n = first_len; }
        else { s = side.data() + side_start; n = side.size() - side_start; }
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
        recs.push_back(r);
        hash.push_back(h);
    };
    size_t pos = lo;
    bool open = false;
    Record cur{};
    const char *first_line = nullptr;
    size_t first_len = 0, lines = 0, side_start = 0;
    while (pos < hi) {
        const char *nl = (const char *)memchr(text + pos, '\n', hi - pos);
        const size_t end = nl ? (size_t)(nl - text) : hi;
        size_t e = end;
        if (e > pos && text[e - 1] == '\r') e--;
        if (e > pos && text[pos] != '#') {
            if (text[pos] == '>') {
                if (open) close_record(cur, first_line, first_len, lines, side_start);
                cur.hdr = text + pos + 1;
                cur.hdr_len = (uint32_t)(e - pos - 1);
                open = true;
                lines = 0; first_line = text + pos; first_len = 0; side_start = side.size();
            } else if (open) {
                if (lines == 0) { first_line = text + pos; first_len = e - pos; }
                else {
                    if (lines == 1) side.append(first_line, first_len);
                    side.append(text + pos, e - pos);
                }
                lines++;
            }
        }
        pos = end + 1;
    }
    if (open) close_record(cur, first_line, first_len, lines, side_start);
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
    std::vector<std::vector<Record>> recs(T);
    std::vector<std::vector<Hash128>> hashes(T);
    FastaScan out;
    out.side.resize(T);
    out.bytes = size;
    team.run([&](unsigned t, unsigned) {
        const size_t lo = cut[t], hi = cut[t + 1];
        recs[t].reserve((hi - lo) / 160 + 16);
        hashes[t].reserve((hi - lo) / 160 + 16);
        if (hi > lo) scan_chunk(text, lo, hi, md5_digest, recs[t], hashes[t], out.side[t]);
    });
    std::vector<size_t> base(T + 1, 0);
    for (unsigned t = 0; t < T; t++) base[t + 1] = base[t] + recs[t].size();
    out.recs.resize(base[T]);
    out.hash.resize(base[T]);
    team.run([&](unsigned t, unsigned) {
        if (recs[t].empty()) return;
        memcpy(out.recs.data() + base[t], recs[t].data(), recs[t].size() * sizeof(Record));
        memcpy(out.hash.data() + base[t], hashes[t].data(), hashes[t].size() * sizeof(Hash128));
    });
    return out;
}

// ------------------------------------------------------------------------------------------------------------------
// dedup: first_of[i] = the first record with record i's sequence; unique reads numbered in file order of their first record; the
// later records of a sequence chained in file order (next_dup)
// ------------------------------------------------------------------------------------------------------------------
struct FastDedup {
    std::vector<uint32_t> uniq_of_rec;  // [n records]
    std::vector<uint32_t> first_rec;    // [n unique]
    std::vector<uint32_t> next_dup;     // [n records] the next record with the same sequence (file order), 0xFFFFFFFF = none
    uint64_t n_dups = 0;
};

inline FastDedup dedup_fast(const FastaScan &sc, Team &team) {
    const size_t n = sc.recs.size();
    if (n >= 0xFFFFFFFFull) throw std::runtime_error("more than 2^32 - 2 reads in one file");
    const unsigned T = team.size();
    constexpr unsigned LOGS = 8, S = 1u << LOGS;  // shards by the hash's top bits
    FastDedup d;
    d.uniq_of_rec.assign(n, 0);
    d.next_dup.assign(n, 0xFFFFFFFFu);
    std::vector<uint32_t> first_of(n);
    // records per (thread, shard), then every shard's index list: thread-major inside a shard = file order
    std::vector<std::vector<uint32_t>> cnt(T, std::vector<uint32_t>(S, 0));
    team.run([&](unsigned t, unsigned) {
        const size_t lo = n * t / T, hi = n * (t + 1) / T;
        for (size_t i = lo; i < hi; i++) cnt[t][sc.hash[i].hi >> (64 - LOGS)]++;
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
    std::vector<uint32_t> order(n);
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
    d.first_rec.resize(firsts[T]);
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

// the unique reads' characters, contiguous (what rk_place_batch takes)
inline void gather_unique(const FastaScan &sc, const FastDedup &d, Team &team, std::vector<char> &seq, std::vector<uint64_t> &off) {
    const size_t n = d.first_rec.size();
    off.assign(n + 1, 0);
    for (size_t i = 0; i < n; i++) off[i + 1] = off[i] + sc.recs[d.first_rec[i]].seq_len;
    seq.resize(off[n] ? off[n] : 1);
    team.run([&](unsigned t, unsigned T) {
        const size_t lo = n * t / T, hi = n * (t + 1) / T;
        for (size_t i = lo; i < hi; i++) {
            const Record &r = sc.recs[d.first_rec[i]];
            memcpy(seq.data() + off[i], r.seq, r.seq_len);
        }
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

struct FastWriteStats {
    uint64_t placed = 0, bytes = 0;
    bool exact_path = false;  // some name needed the whole-document replacements
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
    std::vector<std::string> part(T);
    std::vector<uint64_t> placed(T, 0);
    std::atomic<bool> needs_exact{false};
    team.run([&](unsigned tt, unsigned) {
        const size_t lo = n * tt / T, hi = n * (tt + 1) / T;
        std::string &o = part[tt];
        o.reserve((hi - lo) * 200 + 64);
        char buf[64];
        for (size_t u = lo; u < hi; u++) {
            if (!n_rows[u]) continue;
            // (every placement but the document's first is preceded by the separator the replacements make of "},{")
            o += "\n},{\n\t";
            o += "\"p\":\n\t[";
            for (uint32_t j = 0; j < n_rows[u]; j++) {
                const uint32_t b = branch[u * K + j];
                if (b >= t.nodes.size()) throw std::runtime_error("placement on branch " + std::to_string(b) + " which the tree does not have");
                if (j) o += ",\n\t";
                o.push_back('[');
                const float sv = score[u * K + j];
                const double lv = lwr[u * K + j];
                std::string like, ratio;
                if (std::isfinite(sv)) { auto r = std::to_chars(buf, buf + sizeof(buf) - 1, sv, std::chars_format::scientific); *r.ptr = 0; like = java_layout(buf, (double)sv); } else like = "null";
                if (std::isfinite(lv)) { auto r = std::to_chars(buf, buf + sizeof(buf) - 1, lv, std::chars_format::scientific); *r.ptr = 0; ratio = java_layout(buf, lv); } else ratio = "null";
                if (guppy) { o += distal[b]; o.push_back(','); o += edge[b]; o.push_back(','); o += ratio; o.push_back(','); o += like; }
                else { o += edge[b]; o.push_back(','); o += like; o.push_back(','); o += ratio; o.push_back(','); o += distal[b]; }
                o += ",0.0]";
            }
            o += "],\n\t\"nm\":\n\t[";
            uint32_t rec = d.first_rec[u];
            bool first = true;
            while (rec != 0xFFFFFFFFu) {
                const Record &r = sc.recs[rec];
                size_t len = r.hdr_len;
                if (!first) {
                    const void *sp = memchr(r.hdr, ' ', r.hdr_len);
                    if (sp) len = (size_t)((const char *)sp - r.hdr);
                    o += ",\n\t";
                }
                if (!plain_for_prettifier(r.hdr, len)) needs_exact.store(true, std::memory_order_relaxed);
                o.push_back('[');
                append_jstr(o, r.hdr, len);
                o += ",1]";
                first = false;
                rec = d.next_dup[rec];
            }
            o.push_back(']');
            placed[tt]++;
        }
    });
    FastWriteStats st;
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
    const int fd = open(path.c_str(), O_WRONLY | O_CREAT | O_TRUNC | O_CLOEXEC, 0644);
    if (fd < 0) throw std::runtime_error("cannot write " + path);
    auto put = [&](const char *p, size_t len, uint64_t where) {
        while (len) {
            const ssize_t w = pwrite(fd, p, len, (off_t)where);
            if (w < 0) { if (errno == EINTR) continue; throw std::runtime_error("write to " + path + " failed"); }
            p += w; len -= (size_t)w; where += (uint64_t)w;
        }
    };
    try {
        if (ftruncate(fd, (off_t)total) != 0) throw std::runtime_error("cannot size " + path);
        team.run([&](unsigned tt, unsigned) {
            if (tt == 0) { put(head.data(), head.size(), 0); put(tail.data(), tail.size(), at_off[T]); }
            if (part[tt].empty()) return;
            const size_t skip = tt == first_part ? sep.size() : 0;
            put(part[tt].data() + skip, part[tt].size() - skip, at_off[tt]);
        });
    } catch (...) {
        (void)close(fd);
        throw;
    }
    if (close(fd) != 0) throw std::runtime_error("cannot close " + path);
    st.bytes = total;
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
