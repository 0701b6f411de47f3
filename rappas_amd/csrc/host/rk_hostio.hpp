// rk_hostio.hpp -- native host side around the placement engine, header-only C++17.
//
// What the Java driver does either side of the native call, restated so that `rk_place` (rk_place_main.cpp) runs FASTA +
// `--jsondb` dump -> .jplace without a JVM.  No placement compute happens here.  Reference lines (paths relative to the
// reference root):
//   N3  FASTA ingest  src/inputs/FASTAPointer.java:66-149 ; dedup src/core/algos/PlacementProcess.java:591-629 (MD5, Jacksum)
//   N2  --jsondb      src/main_v2/SessionNext_v2.java:214-270 (json-simple dump with two bare toString() tokens)
//       --uniondb     src/main_v2/SessionNext_v2.java:109-207 (Java serialization stream, read by rk_javaser.hpp)
//   N1  jplace        src/main_v2/Main_PLACEMENT_v07.java:224-315, src/core/algos/PlacementProcess.java:1005-1046,
//                     src/tree/NewickReader.java:46-160, src/tree/PhyloTree.java:408-439, src/tree/NewickWriter.java:116-212
// rappas_amd/hostio.py is the same logic in Python; tests/test_host_cpp.py requires byte-identical output from the two.
#pragma once
#include <array>
#include <charconv>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <thread>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "rk_javaser.hpp"

namespace rkh {

// ------------------------------------------------------------------------------------------------------------------
// MD5 (RFC 1321) -- the checksum PlacementProcess.java:505-510 asks Jacksum for
// ------------------------------------------------------------------------------------------------------------------
inline std::array<uint8_t, 16> md5(const std::string &msg) {
    static const uint32_t K[64] = {
        0xd76aa478, 0xe8c7b756, 0x242070db, 0xc1bdceee, 0xf57c0faf, 0x4787c62a, 0xa8304613, 0xfd469501, 0x698098d8, 0x8b44f7af,
        0xffff5bb1, 0x895cd7be, 0x6b901122, 0xfd987193, 0xa679438e, 0x49b40821, 0xf61e2562, 0xc040b340, 0x265e5a51, 0xe9b6c7aa,
        0xd62f105d, 0x02441453, 0xd8a1e681, 0xe7d3fbc8, 0x21e1cde6, 0xc33707d6, 0xf4d50d87, 0x455a14ed, 0xa9e3e905, 0xfcefa3f8,
        0x676f02d9, 0x8d2a4c8a, 0xfffa3942, 0x8771f681, 0x6d9d6122, 0xfde5380c, 0xa4beea44, 0x4bdecfa9, 0xf6bb4b60, 0xbebfbc70,
        0x289b7ec6, 0xeaa127fa, 0xd4ef3085, 0x04881d05, 0xd9d4d039, 0xe6db99e5, 0x1fa27cf8, 0xc4ac5665, 0xf4292244, 0x432aff97,
        0xab9423a7, 0xfc93a039, 0x655b59c3, 0x8f0ccc92, 0xffeff47d, 0x85845dd1, 0x6fa87e4f, 0xfe2ce6e0, 0xa3014314, 0x4e0811a1,
        0xf7537e82, 0xbd3af235, 0x2ad7d2bb, 0xeb86d391};
    static const int R[64] = {7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 5, 9,  14, 20, 5, 9,
                              14, 20, 5, 9,  14, 20, 5, 9,  14, 20, 4, 11, 16, 23, 4, 11, 16, 23, 4, 11, 16, 23,
                              4, 11, 16, 23, 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21};
    uint32_t h0 = 0x67452301, h1 = 0xefcdab89, h2 = 0x98badcfe, h3 = 0x10325476;
    auto block = [&](const unsigned char *blk) {
        uint32_t w[16];
        for (int i = 0; i < 16; i++) {
            const unsigned char *p = blk + 4 * i;
            w[i] = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
        }
        uint32_t a = h0, b = h1, c = h2, d = h3;
        for (int i = 0; i < 64; i++) {
            uint32_t f;
            int g;
            if (i < 16) { f = (b & c) | (~b & d); g = i; }
            else if (i < 32) { f = (d & b) | (~d & c); g = (5 * i + 1) % 16; }
            else if (i < 48) { f = b ^ c ^ d; g = (3 * i + 5) % 16; }
            else { f = c ^ (b | ~d); g = (7 * i) % 16; }
            const uint32_t tmp = d;
            d = c;
            c = b;
            const uint32_t x = a + f + K[i] + w[g];
            b = b + ((x << R[i]) | (x >> (32 - R[i])));
            a = tmp;
        }
        h0 += a; h1 += b; h2 += c; h3 += d;
    };
    // whole 64-byte blocks straight from the message, then the padded tail (0x80, zeros, 64-bit bit length) from a small buffer
    const unsigned char *data = (const unsigned char *)msg.data();
    const size_t n = msg.size();
    size_t off = 0;
    for (; off + 64 <= n; off += 64) block(data + off);
    unsigned char tail[128] = {0};
    const size_t rem = n - off;
    memcpy(tail, data + off, rem);
    tail[rem] = 0x80;
    const size_t tail_len = rem < 56 ? 64 : 128;
    const uint64_t bitlen = (uint64_t)n * 8;
    for (int i = 0; i < 8; i++) tail[tail_len - 8 + i] = (unsigned char)((bitlen >> (8 * i)) & 0xFF);
    block(tail);
    if (tail_len == 128) block(tail + 64);
    std::array<uint8_t, 16> out;
    const uint32_t hs[4] = {h0, h1, h2, h3};
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) out[4 * i + j] = (uint8_t)((hs[i] >> (8 * j)) & 0xFF);
    return out;
}

// ------------------------------------------------------------------------------------------------------------------
// N3: FASTA + dedup
// ------------------------------------------------------------------------------------------------------------------
struct Fasta {
    std::string header, seq;
};

inline std::string trim(const std::string &s) {  // String.trim(): characters <= ' '
    size_t a = 0, b = s.size();
    while (a < b && (unsigned char)s[a] <= ' ') a++;
    while (b > a && (unsigned char)s[b - 1] <= ' ') b--;
    return s.substr(a, b - a);
}

// FASTAPointer.nextSequenceAsFasta: empty lines and '#' lines skipped, '>' opens a record, lines concatenated, trim()
inline std::vector<Fasta> read_fasta(const std::string &text) {
    std::vector<Fasta> out;
    bool open = false;
    Fasta cur;
    size_t pos = 0;
    while (pos <= text.size()) {
        size_t nl = text.find('\n', pos);
        if (nl == std::string::npos) nl = text.size();
        std::string line = text.substr(pos, nl - pos);
        if (!line.empty() && line.back() == '\r') line.pop_back();  // (str.splitlines() of the Python twin)
        pos = nl + 1;
        if (line.empty() || line[0] == '#') { if (nl == text.size()) break; continue; }
        if (line[0] == '>') {
            if (open) { cur.seq = trim(cur.seq); out.push_back(cur); }
            cur = Fasta{line.substr(1), ""};
            open = true;
        } else if (open) {
            cur.seq += line;
        }
        if (nl == text.size()) break;
    }
    if (open) { cur.seq = trim(cur.seq); out.push_back(cur); }
    return out;
}

// PlacementProcess.java:591-629: first occurrence keeps its FULL header, later duplicates the header cut at the first space.
// The checksums (MD5 of the sequence without '-') are independent of each other: they are computed on a few host threads,
// then the reads are walked in file order through a hash map keyed by the digest (its first 8 bytes are hash enough).
inline std::array<uint8_t, 16> read_checksum(const Fasta &f) {
    if (f.seq.find('-') == std::string::npos) return md5(f.seq);
    std::string nogap;
    nogap.reserve(f.seq.size());
    for (char c : f.seq)
        if (c != '-') nogap.push_back(c);
    return md5(nogap);
}

struct DigestHash {
    size_t operator()(const std::array<uint8_t, 16> &d) const {
        uint64_t v;
        memcpy(&v, d.data(), 8);
        return (size_t)v;
    }
};

// index form of the dedup: which unique read every record belongs to, and the record that introduced each unique read
struct Dedup {
    std::vector<uint32_t> uniq_of_rec;   // [n records]
    std::vector<uint32_t> first_rec;     // [n unique]
};

inline Dedup dedup_index(const std::vector<Fasta> &recs, unsigned n_threads = 0) {
    const size_t n = recs.size();
    std::vector<std::array<uint8_t, 16>> digest(n);
    unsigned hw = std::thread::hardware_concurrency();
    unsigned T = n_threads ? n_threads : std::max(1u, std::min(hw ? hw : 1u, 16u));
    if (n < 4096) T = 1;
    auto work = [&](size_t lo, size_t hi) { for (size_t i = lo; i < hi; i++) digest[i] = read_checksum(recs[i]); };
    std::vector<std::thread> th;
    for (unsigned t = 1; t < T; t++) th.emplace_back(work, n * t / T, n * (t + 1) / T);
    work(0, n / T);
    for (std::thread &x : th) x.join();
    Dedup d;
    d.uniq_of_rec.resize(n);
    std::unordered_map<std::array<uint8_t, 16>, uint32_t, DigestHash> index;
    index.reserve(n);
    for (size_t i = 0; i < n; i++) {
        auto ins = index.emplace(digest[i], (uint32_t)d.first_rec.size());
        if (ins.second) d.first_rec.push_back((uint32_t)i);
        d.uniq_of_rec[i] = ins.first->second;
    }
    return d;
}

// the "nm" lists of the jplace records: the first occurrence's full header, then the space-cut headers of its duplicates
inline std::vector<std::vector<std::string>> dedup_names(const std::vector<Fasta> &recs, const Dedup &d) {
    std::vector<std::vector<std::string>> names(d.first_rec.size());
    for (size_t i = 0; i < recs.size(); i++) {
        const uint32_t u = d.uniq_of_rec[i];
        if (d.first_rec[u] == i) { names[u].insert(names[u].begin(), recs[i].header); continue; }
        const size_t cut = recs[i].header.find(' ');
        names[u].push_back(cut == std::string::npos ? recs[i].header : recs[i].header.substr(0, cut));
    }
    return names;
}

inline void dedup_reads(const std::vector<Fasta> &recs, std::vector<Fasta> &unique, std::vector<std::vector<std::string>> &names,
                        unsigned n_threads = 0) {
    const Dedup d = dedup_index(recs, n_threads);
    for (uint32_t r : d.first_rec) unique.push_back(recs[r]);
    names = dedup_names(recs, d);
}

// notplaced_<query>.tsv (Main_PLACEMENT_v07.java:214, PlacementProcess.java:797-806): the full header of every read that hits
// nothing in the database, every occurrence (the reference registers checksums of placed reads only, :1046), in file order
inline std::string notplaced_log(const std::vector<Fasta> &recs, const Dedup &d, const uint32_t *flags) {
    std::string out;
    for (size_t i = 0; i < recs.size(); i++)
        if (!(flags[d.uniq_of_rec[i]] & 1u)) { out += recs[i].header; out += "\n"; }
    return out;
}

// ------------------------------------------------------------------------------------------------------------------
// N1: tree
// ------------------------------------------------------------------------------------------------------------------
struct Node {
    int id = 0;
    std::string label;
    float bl = 0.0f;
    std::vector<int> children;
    int parent = -1;
    int jplace_edge = -1;
};

struct Tree {
    std::vector<Node> nodes;  // by id (order of appearance, root = 0)
    int root = 0;
    bool rooted() const { return nodes[root].children.size() == 2; }  // NewickReader.java:209-220
};

inline void fill_node(Node &n, const std::string &text) {
    if (text.empty()) return;
    const size_t c = text.find(':');
    n.label = text.substr(0, c);
    if (c != std::string::npos) {
        std::string rest = text.substr(c + 1);
        const size_t c2 = rest.find(':');
        if (c2 != std::string::npos) rest = rest.substr(0, c2);
        const size_t br = rest.find('{');
        if (br != std::string::npos) rest = rest.substr(0, br);
        if (!rest.empty()) n.bl = (float)strtod(rest.c_str(), nullptr);
    }
}

inline void reset_jplace_edge_ids(Tree &t) {  // PhyloTree.java:408-439
    int counter = -1;
    struct F { int node; size_t next; };
    std::vector<F> st{{t.root, 0}};
    while (!st.empty()) {
        F &f = st.back();
        Node &n = t.nodes[f.node];
        if (f.next < n.children.size()) {
            const int c = n.children[f.next++];
            if (t.nodes[c].children.empty()) t.nodes[c].jplace_edge = ++counter;
            else st.push_back({c, 0});
        } else {
            n.jplace_edge = ++counter;
            st.pop_back();
        }
    }
}

inline Tree parse_newick(const std::string &text) {  // NewickReader.java:46-200: ids in order of appearance
    Tree t;
    const std::string s = trim(text);
    std::vector<int> stack;
    std::vector<std::vector<int>> cur(1);
    std::string token;
    int last_closed = -1;
    auto new_node = [&]() { Node n; n.id = (int)t.nodes.size(); t.nodes.push_back(n); return n.id; };
    for (char ch : s) {
        if (ch == '(') {
            stack.push_back(new_node());
            cur.emplace_back();
            token.clear();
            last_closed = -1;
        } else if (ch == ',' || ch == ')' || ch == ';') {
            const std::string txt = trim(token);
            token.clear();
            if (last_closed >= 0) {
                fill_node(t.nodes[last_closed], txt);
                last_closed = -1;
            } else if (!txt.empty() || ch != ';') {
                const int leaf = new_node();
                fill_node(t.nodes[leaf], txt);
                cur.back().push_back(leaf);
            }
            if (ch == ')') {
                if (stack.empty() || cur.size() < 2) throw std::runtime_error("newick: unbalanced ')'");
                const int parent = stack.back();
                stack.pop_back();
                for (int c : cur.back()) { t.nodes[c].parent = parent; t.nodes[parent].children.push_back(c); }
                cur.pop_back();
                cur.back().push_back(parent);
                last_closed = parent;
            } else if (ch == ';') {
                break;
            }
        } else {
            token.push_back(ch);
        }
    }
    if (cur.empty() || cur[0].empty()) throw std::runtime_error("newick: no tree found");
    t.root = cur[0][0];
    reset_jplace_edge_ids(t);
    return t;
}

// NumberFormat.getNumberInstance(Locale.UK), exactly 12 fraction digits, grouping commas (NewickWriter.java:61-64)
inline std::string fmt12(float x) {
    char buf[400];
    snprintf(buf, sizeof(buf), "%.12f", (double)x);  // glibc: exact binary value, ties to even
    std::string s(buf);
    const bool neg = !s.empty() && s[0] == '-';
    if (neg) s.erase(0, 1);
    const size_t dot = s.find('.');
    std::string ip = s.substr(0, dot), out;
    for (size_t i = 0; i < ip.size(); i++) {
        out.push_back(ip[i]);
        const size_t left = ip.size() - 1 - i;
        if (left && left % 3 == 0) out.push_back(',');
    }
    return (neg ? "-" : "") + out + s.substr(dot);
}

inline void newick_dfs(const Tree &t, int id, int level, bool bl, bool names, bool jlabels, std::string &out) {
    const Node &node = t.nodes[id];
    out.push_back('(');
    const size_t n = node.children.size();
    for (size_t i = 0; i < n; i++) {
        const Node &c = t.nodes[node.children[i]];
        if (c.children.empty()) {
            out += c.label;
            if (bl) out += ":" + fmt12(c.bl);
            if (jlabels) out += "{" + std::to_string(c.jplace_edge) + "}";
        } else {
            newick_dfs(t, c.id, level + 1, bl, names, jlabels, out);
        }
        if (i + 1 < n) {
            out.push_back(',');
        } else {
            out.push_back(')');
            if (names) out += node.label;
            if (bl && level > -1) out += ":" + fmt12(node.bl);
            if (jlabels && level > -1) out += "{" + std::to_string(node.jplace_edge) + "}";
        }
    }
    if (node.parent < 0) out.push_back(';');
}

inline std::string write_newick(const Tree &t, bool bl, bool names, bool jlabels) {  // NewickWriter.java:116-212
    std::string out;
    newick_dfs(t, t.root, t.rooted() ? 0 : -1, bl, names, jlabels, out);
    return out;
}
inline std::string jplace_newick(const Tree &t) { return write_newick(t, true, true, true); }

// ------------------------------------------------------------------------------------------------------------------
// N1: numbers the way json-simple prints them (Number.toString()): shortest digits, Java's layout
// ------------------------------------------------------------------------------------------------------------------
inline std::string java_layout(const char *sci, double value) {
    // sci = std::to_chars scientific output: [-]d[.ddd]e[+-]XX
    std::string s(sci);
    bool neg = false;
    if (!s.empty() && s[0] == '-') { neg = true; s.erase(0, 1); }
    const size_t e = s.find('e');
    std::string mant = s.substr(0, e);
    const int exp10 = atoi(s.c_str() + e + 1);
    std::string ds;
    for (char c : mant)
        if (c != '.') ds.push_back(c);
    while (ds.size() > 1 && ds.back() == '0') ds.pop_back();
    if (value == 0) return std::string(std::signbit(value) ? "-" : "") + "0.0";
    const int e10 = exp10 + 1;  // value = 0.ds * 10^e10
    const double a = std::fabs(value);
    std::string body;
    if (a >= 1e-3 && a < 1e7) {
        if (e10 <= 0) body = "0." + std::string((size_t)(-e10), '0') + ds;
        else if ((size_t)e10 >= ds.size()) body = ds + std::string((size_t)e10 - ds.size(), '0') + ".0";
        else body = ds.substr(0, (size_t)e10) + "." + ds.substr((size_t)e10);
    } else {
        body = ds.substr(0, 1) + "." + (ds.size() > 1 ? ds.substr(1) : std::string("0")) + "E" + std::to_string(e10 - 1);
    }
    return (neg ? "-" : "") + body;
}
inline std::string java_float_to_string(float x) {
    if (!std::isfinite(x)) return "null";  // JSONValue.toJSONString: non-finite numbers become null
    char buf[64];
    auto r = std::to_chars(buf, buf + sizeof(buf) - 1, x, std::chars_format::scientific);
    *r.ptr = 0;
    return java_layout(buf, (double)x);
}
inline std::string java_double_to_string(double x) {
    if (!std::isfinite(x)) return "null";
    char buf[64];
    auto r = std::to_chars(buf, buf + sizeof(buf) - 1, x, std::chars_format::scientific);
    *r.ptr = 0;
    return java_layout(buf, x);
}

// JSONValue.escape of json-simple 1.1 (the reference's lib/json_simple-1.1.jar): the short escapes incl. "\/", and \uXXXX
// (upper-case hex) for U+0000-001F, U+007F-009F and U+2000-20FF; everything else verbatim.  Input and output are UTF-8.
inline std::string jstr(const std::string &s) {
    std::string o = "\"";
    auto esc = [&](unsigned cp) { char b[8]; snprintf(b, sizeof(b), "\\u%04X", cp); o += b; };
    for (size_t i = 0; i < s.size(); i++) {
        const unsigned char c = (unsigned char)s[i];
        switch (c) {
        case '"': o += "\\\""; continue;
        case '\\': o += "\\\\"; continue;
        case '\b': o += "\\b"; continue;
        case '\f': o += "\\f"; continue;
        case '\n': o += "\\n"; continue;
        case '\r': o += "\\r"; continue;
        case '\t': o += "\\t"; continue;
        case '/': o += "\\/"; continue;
        default: break;
        }
        if (c <= 0x1F || c == 0x7F) { esc(c); continue; }
        if (c == 0xC2 && i + 1 < s.size() && (unsigned char)s[i + 1] >= 0x80 && (unsigned char)s[i + 1] <= 0x9F) {  // U+0080-009F
            esc((unsigned char)s[i + 1]);
            i += 1;
            continue;
        }
        if (c == 0xE2 && i + 2 < s.size() && (unsigned char)s[i + 1] >= 0x80 && (unsigned char)s[i + 1] <= 0x83) {  // U+2000-20FF
            esc(0x2000u + (((unsigned char)s[i + 1] & 0x3Fu) << 6) + ((unsigned char)s[i + 2] & 0x3Fu));
            i += 2;
            continue;
        }
        o.push_back((char)c);
    }
    return o + "\"";
}

inline void replace_all(std::string &s, const std::string &from, const std::string &to) {
    std::string out;
    size_t pos = 0, hit;
    while ((hit = s.find(from, pos)) != std::string::npos) {
        out.append(s, pos, hit - pos);
        out += to;
        pos = hit + from.size();
    }
    out.append(s, pos, std::string::npos);
    s.swap(out);
}

struct Placement {
    std::vector<std::array<std::string, 5>> rows;
    std::vector<std::string> names;
};

// PlacementProcess.java:1005-1046
inline std::vector<Placement> jplace_placements(const Tree &t, const std::vector<std::vector<std::string>> &names, size_t n,
                                                uint32_t K, const uint8_t *n_rows, const uint16_t *branch, const float *score,
                                                const double *lwr, bool guppy) {
    std::vector<Placement> out;
    for (size_t i = 0; i < n; i++) {
        if (!n_rows[i]) continue;
        Placement p;
        for (uint32_t j = 0; j < n_rows[i]; j++) {
            const uint32_t b = branch[i * K + j];
            if (b >= t.nodes.size()) throw std::runtime_error("placement on branch " + std::to_string(b) + " which the tree does not have");
            const Node &node = t.nodes[b];
            const std::string edge = std::to_string(node.jplace_edge), like = java_float_to_string(score[i * K + j]),
                              ratio = java_double_to_string(lwr[i * K + j]), distal = java_float_to_string(node.bl / 2.0f);
            if (guppy) p.rows.push_back({distal, edge, ratio, like, "0.0"});
            else p.rows.push_back({edge, like, ratio, distal, "0.0"});
        }
        p.names = names[i];
        out.push_back(std::move(p));
    }
    return out;
}

// Main_PLACEMENT_v07.java:224-315: HashMap key order metadata, tree, placements, fields, version / p, nm; regex prettifier
inline std::string jplace_document(const Tree &t, const std::vector<Placement> &pl, const std::string &call_string, bool guppy) {
    const char *f_std[5] = {"edge_num", "likelihood", "like_weight_ratio", "distal_length", "pendant_length"};
    const char *f_gup[5] = {"distal_length", "edge_num", "like_weight_ratio", "likelihood", "pendant_length"};
    std::string out = "{\"metadata\":{\"invocation\":" + jstr("viromeplacer" + call_string) + "},\"tree\":" + jstr(jplace_newick(t)) +
                      ",\"placements\":[";
    for (size_t i = 0; i < pl.size(); i++) {
        if (i) out.push_back(',');
        out += "{\"p\":[";
        for (size_t r = 0; r < pl[i].rows.size(); r++) {
            if (r) out.push_back(',');
            out.push_back('[');
            for (int c = 0; c < 5; c++) { if (c) out.push_back(','); out += pl[i].rows[r][c]; }
            out.push_back(']');
        }
        out += "],\"nm\":[";
        for (size_t q = 0; q < pl[i].names.size(); q++) {
            if (q) out.push_back(',');
            out += "[" + jstr(pl[i].names[q]) + ",1]";
        }
        out += "]}";
    }
    out += "],\"fields\":[";
    for (int c = 0; c < 5; c++) { if (c) out.push_back(','); out += jstr(guppy ? f_gup[c] : f_std[c]); }
    out += "],\"version\":3}";
    replace_all(out, "},{", "\n},{\n\t");
    replace_all(out, "],\"", "],\n\t\"");
    replace_all(out, "]}],", "]\n}\n],\n");
    replace_all(out, ",\"placements\":[{\"p\"", ",\n\"placements\":\n[\n{\n\t\"p\"");
    replace_all(out, "],[", "],\n\t[");
    replace_all(out, "\"p\":[[", "\"p\":\n\t[[");
    replace_all(out, "\"nm\":[[", "\"nm\":\n\t[[");
    return out;
}

// ------------------------------------------------------------------------------------------------------------------
// N2: --jsondb dump (SessionNext_v2.saveToJSON), tolerant of the bare `states` / `align` tokens
// ------------------------------------------------------------------------------------------------------------------
struct JsonDb {
    uint32_t k = 0;
    float thr = 0, thr_log10 = 0;
    std::string original_tree;
    std::vector<uint64_t> key_codes, row_offsets{0};
    std::vector<uint16_t> branch_ids;
    std::vector<float> scores;
};

class JsonScan {
  public:
    explicit JsonScan(const std::string &s) : s_(s) {}
    void ws() { while (p_ < s_.size() && (unsigned char)s_[p_] <= ' ') p_++; }
    char peek() { ws(); return p_ < s_.size() ? s_[p_] : '\0'; }
    bool eat(char c) { if (peek() == c) { p_++; return true; } return false; }
    void need(char c) { if (!eat(c)) fail(std::string("expected '") + c + "'"); }
    [[noreturn]] void fail(const std::string &m) { throw std::runtime_error("jsondb: " + m + " at byte " + std::to_string(p_)); }
    std::string str() {
        need('"');
        std::string o;
        while (p_ < s_.size() && s_[p_] != '"') {
            char c = s_[p_++];
            if (c == '\\' && p_ < s_.size()) {
                const char e = s_[p_++];
                switch (e) {
                case 'n': o.push_back('\n'); break;
                case 't': o.push_back('\t'); break;
                case 'r': o.push_back('\r'); break;
                case 'b': o.push_back('\b'); break;
                case 'f': o.push_back('\f'); break;
                case 'u': {
                    if (p_ + 4 > s_.size()) fail("short \\u escape");
                    const unsigned cp = (unsigned)strtoul(s_.substr(p_, 4).c_str(), nullptr, 16);
                    p_ += 4;
                    if (cp < 0x80) o.push_back((char)cp);
                    else if (cp < 0x800) { o.push_back((char)(0xC0 | (cp >> 6))); o.push_back((char)(0x80 | (cp & 0x3F))); }
                    else { o.push_back((char)(0xE0 | (cp >> 12))); o.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); o.push_back((char)(0x80 | (cp & 0x3F))); }
                    break;
                }
                default: o.push_back(e);
                }
            } else {
                o.push_back(c);
            }
        }
        if (p_ >= s_.size()) fail("unterminated string");
        p_++;
        return o;
    }
    std::string scalar() {  // a number, true/false/null, or one of the reference's bare object tokens: up to ',' or '}'
        ws();
        const size_t a = p_;
        while (p_ < s_.size() && s_[p_] != ',' && s_[p_] != '}' && s_[p_] != ']') p_++;
        return trim(s_.substr(a, p_ - a));
    }
    void skip_value() {
        const char c = peek();
        if (c == '"') { str(); return; }
        if (c == '{' || c == '[') {
            const char close = c == '{' ? '}' : ']';
            p_++;
            if (eat(close)) return;
            do {
                if (c == '{') { str(); need(':'); }
                skip_value();
            } while (eat(','));
            need(close);
            return;
        }
        scalar();
    }

  private:
    const std::string &s_;
    size_t p_ = 0;
};

inline JsonDb load_jsondb(const std::string &text) {
    JsonDb db;
    JsonScan js(text);
    js.need('{');
    bool have_hash = false;
    if (!js.eat('}')) {
        do {
            const std::string key = js.str();
            js.need(':');
            if (key == "k") db.k = (uint32_t)atoi(js.scalar().c_str());
            else if (key == "PPStarThreshold") db.thr = strtof(js.scalar().c_str(), nullptr);
            else if (key == "PPStarThresholdAsLog10") db.thr_log10 = strtof(js.scalar().c_str(), nullptr);
            else if (key == "originalTree") db.original_tree = js.str();
            else if (key == "hash") {
                have_hash = true;
                js.need('{');
                if (!js.eat('}')) {
                    do {
                        const std::string kmer = js.str();
                        uint64_t code = 0;
                        for (size_t i = 0; i < kmer.size(); i++) {
                            int st;
                            switch (kmer[i]) {  // DNAStatesShifted: A=0 T=1 C=2 G=3, base i at bits 2i
                            case 'A': st = 0; break;
                            case 'T': st = 1; break;
                            case 'C': st = 2; break;
                            case 'G': st = 3; break;
                            default: throw std::runtime_error("jsondb: k-mer \"" + kmer + "\" is not a DNA k-mer (amino-acid dumps are unusable: AAStates.expandMer ignores its argument)");
                            }
                            code |= (uint64_t)st << (2 * i);
                        }
                        db.key_codes.push_back(code);
                        js.need(':');
                        js.need('{');
                        if (!js.eat('}')) {
                            do {
                                const std::string node = js.str();
                                js.need(':');
                                db.branch_ids.push_back((uint16_t)atoi(node.c_str()));
                                db.scores.push_back(strtof(js.scalar().c_str(), nullptr));
                            } while (js.eat(','));
                            js.need('}');
                        }
                        db.row_offsets.push_back(db.branch_ids.size());
                        if (db.k && kmer.size() != db.k) throw std::runtime_error("jsondb: k-mer \"" + kmer + "\" is not a DNA " + std::to_string(db.k) + "-mer");
                    } while (js.eat(','));
                    js.need('}');
                }
            } else {
                js.skip_value();
            }
        } while (js.eat(','));
        js.need('}');
    }
    if (!db.k || !have_hash || db.original_tree.empty()) throw std::runtime_error("jsondb: k, originalTree or hash missing");
    for (uint64_t c : db.key_codes)
        if (db.k < 32 && (c >> (2 * db.k))) throw std::runtime_error("jsondb: k-mer longer than k");
    return db;
}

// ------------------------------------------------------------------------------------------------------------------
// N2: a `.union` database (SessionNext_v2.storeHash, src/main_v2/SessionNext_v2.java:109-147; read back by load, :158-207): a
// Java serialization stream holding, in this order, block data {int k, int minK, float omega, int branchPerEdge, float
// stateThreshold, float PPStarThreshold, float PPStarThresholdAsLog10}, the objects states, align, originalTree, extendedTree,
// ARTree, nodeMapping, block data {float calibrationNormScore, boolean onlyFakes} and the CustomHash_v4_FastUtil81.  Taken from
// the object graph (the same as rappas_amd/hostio.py: load_uniondb):
//   * alphabet: the class of `states` (core.DNAStatesShifted / core.AAStates; --convertUO shows as 'U' in AAStates' char map);
//   * tree: PhyloTree.indexById (HashMap<Integer, PhyloNode>, src/tree/PhyloTree.java:39) -> per node id, label, branch length,
//     jplace edge id (src/tree/PhyloNode.java:30-37) and, from its DefaultMutableTreeNode part, parent and ordered children;
//   * rows: CustomHash_v4_FastUtil81.hash, an Object2ObjectOpenCustomHashMap<byte[], Char2FloatOpenHashMap>
//     (src/core/hash/CustomHash_v4_FastUtil81.java:36): fastutil writes its open-hash maps as defaultWriteObject() followed by
//     the entries -- writeObject(key), writeObject(value) for the outer map, writeChar(key), writeFloat(value) for a row.  DNA
//     keys are compressMer bytes (DNAStatesShifted.java:115-143: little-endian 2-bit codes), AA keys one state per byte.
// PARITY UNPINNED (rk_javaser.hpp): never run against a file written by a JVM.
// ------------------------------------------------------------------------------------------------------------------
struct UnionDb {
    uint32_t alphabet = 0, k = 0;
    bool convert_uo = false, only_fakes = false;
    float thr = 0, thr_log10 = 0, omega = 0, calibration = 0;
    Tree tree;
    std::vector<uint64_t> key_codes, row_offsets{0};
    std::vector<uint16_t> branch_ids;
    std::vector<float> scores;
};

inline UnionDb load_uniondb(const std::string &data) {
    using rkjs::Node;
    using rkjs::P;
    auto be32 = [](const char *b) { return (uint32_t)(uint8_t)b[0] << 24 | (uint32_t)(uint8_t)b[1] << 16 | (uint32_t)(uint8_t)b[2] << 8 | (uint32_t)(uint8_t)b[3]; };
    auto bef = [&](const char *b) { const uint32_t u = be32(b); float x; memcpy(&x, &u, 4); return x; };
    // Record by record: only states (object 0), originalTree (2) and the hash (6) are kept -- the alignment, the extended tree, the AR
    // tree and the node mapping are read (the stream has to be) and let go of at once; and the hash's rows are turned into CSR
    // as they are read (rk_javaser.hpp: stream_annotations_of), never held as objects: a 10^7-row session loads in the memory of its
    // CSR form plus the file (round 3: an object graph of tens of GB).
    UnionDb db;
    const std::string oname = "it.unimi.dsi.fastutil.objects.Object2ObjectOpenCustomHashMap", rname = "it.unimi.dsi.fastutil.chars.Char2FloatOpenHashMap";
    uint32_t alphabet_seen = 0;
    rkjs::Reader reader((const uint8_t *)data.data(), data.size());
    P key_pending;
    bool have_key = false;
    uint64_t rows_seen = 0;
    reader.stream_annotations_of(oname, [&](const P &x) {
        if (!have_key) { key_pending = x; have_key = true; return; }
        const P key = key_pending, row = x;
        key_pending.reset();
        have_key = false;
        if (!key || key->kind != Node::ARRAY || !row || row->kind != Node::OBJECT) throw std::runtime_error("union: malformed entry of the outer map");
        if (alphabet_seen == 0) throw std::runtime_error("union: the hash precedes the states object");
        const std::string &raw = key->s;
        uint64_t code = 0;
        if (alphabet_seen == 4) {
            if (raw.size() > 8) throw std::runtime_error("union: DNA key longer than 8 bytes");
            for (size_t b = 0; b < raw.size(); b++) code |= (uint64_t)(uint8_t)raw[b] << (8 * b);  // compressMer bytes: base i at bits 2*(i%4) of byte i/4
            if (db.k < 32 && (code >> (2 * db.k))) throw std::runtime_error("union: DNA key has bits beyond 2k");
        } else {
            if (raw.size() > 12) throw std::runtime_error("union: amino-acid key longer than 12 states");
            for (size_t b = 0; b < raw.size(); b++) code |= (uint64_t)(uint8_t)raw[b] << (5 * b);
        }
        db.key_codes.push_back(code);
        const P rsize = row->get("size");
        const int64_t m = rsize ? rsize->i : -1;
        const std::string blob = row->block(rname);
        if (m < 0 || blob.size() != (size_t)(6 * m))
            throw std::runtime_error("union: row announces " + std::to_string(m) + " entries, stream holds " + std::to_string(blob.size()) + " bytes");
        for (int64_t e = 0; e < m; e++) {
            const char *b = &blob[(size_t)(6 * e)];
            db.branch_ids.push_back((uint16_t)((uint8_t)b[0] << 8 | (uint8_t)b[1]));
            db.scores.push_back(bef(b + 2));
        }
        db.row_offsets.push_back(db.branch_ids.size());
        rows_seen++;
    });
    std::string blocks;
    std::vector<P> objs;
    reader.contents([&](size_t, rkjs::Record &&r) {
        if (r.is_block) {
            blocks += r.value->s;
            if (blocks.size() >= 4 && db.k == 0) db.k = be32(&blocks[0]);  // (k precedes every object: the keys' range check needs it)
            return true;
        }
        const size_t oi = objs.size();
        const bool keep = oi == 0 || oi == 2 || oi == 6;
        if (oi == 0 && r.value && r.value->kind == Node::OBJECT)
            alphabet_seen = r.value->classname() == "core.DNAStatesShifted" ? 4u : (r.value->classname() == "core.AAStates" ? 20u : 0u);
        objs.push_back(keep ? r.value : nullptr);
        return keep;
    });
    if (blocks.size() < 33 || objs.size() < 7)
        throw std::runtime_error("union: expected 33 bytes of scalars and 7 objects, found " + std::to_string(blocks.size()) + " and " +
                                 std::to_string(objs.size()) + " (a database stored without its hash?)");
    db.k = be32(&blocks[0]);
    db.omega = bef(&blocks[8]);
    db.thr = bef(&blocks[20]);
    db.thr_log10 = bef(&blocks[24]);
    db.calibration = bef(&blocks[28]);
    db.only_fakes = blocks[32] != 0;
    const P states = objs[0], otree = objs[2], chash = objs[6];
    if (!states || states->kind != Node::OBJECT || !otree || otree->kind != Node::OBJECT || !chash || chash->kind != Node::OBJECT)
        throw std::runtime_error("union: states, originalTree or the hash is not an object");
    if (states->classname() == "core.DNAStatesShifted") {
        db.alphabet = 4;
    } else if (states->classname() == "core.AAStates") {
        db.alphabet = 20;
        if (const P b = states->get("b"))
            for (const auto &kv : rkjs::hashmap_items(*b))
                if (kv.first && rkjs::boxed_int(kv.first) == 'U') db.convert_uo = true;
    } else {
        throw std::runtime_error("union: unknown States class " + states->classname());
    }
    // ---- original tree ----
    const P index = otree->get("indexById");
    if (!index) throw std::runtime_error("union: originalTree has no indexById map");
    const auto items = rkjs::hashmap_items(*index);
    const size_t n_nodes = items.size();
    db.tree.nodes.assign(n_nodes, rkh::Node());
    std::map<const Node *, int> ident;
    std::vector<P> jn(n_nodes);
    std::vector<bool> seen(n_nodes, false);
    for (const auto &kv : items) {
        const int64_t id = rkjs::boxed_int(kv.first);
        if (id < 0 || (size_t)id >= n_nodes) throw std::runtime_error("union: node id " + std::to_string(id) + " outside 0.." + std::to_string(n_nodes - 1));
        if (!kv.second || kv.second->kind != Node::OBJECT || seen[(size_t)id]) throw std::runtime_error("union: indexById holds node " + std::to_string(id) + " twice or as null");
        seen[(size_t)id] = true;
        jn[(size_t)id] = kv.second;
        ident[kv.second.get()] = (int)id;
        rkh::Node &n = db.tree.nodes[(size_t)id];
        n.id = (int)id;
        const P label = kv.second->get("label"), bl = kv.second->get("branchLengthToAncestor"), je = kv.second->get("jplaceEdgeId");
        n.label = label && label->kind == Node::STRING ? label->s : std::string();
        if (!bl || !je) throw std::runtime_error("union: PhyloNode without branchLengthToAncestor / jplaceEdgeId");
        n.bl = (float)bl->f;
        n.jplace_edge = (int)je->i;
    }
    for (size_t id = 0; id < n_nodes; id++) {
        const P ch = jn[id]->get("children");
        if (!ch) continue;
        for (const P &c : rkjs::list_items(*ch)) {
            auto it = ident.find(c.get());
            if (it == ident.end()) throw std::runtime_error("union: a child of node " + std::to_string(id) + " is not in indexById");
            db.tree.nodes[id].children.push_back(it->second);
            db.tree.nodes[(size_t)it->second].parent = (int)id;
        }
    }
    db.tree.root = -1;
    for (size_t id = 0; id < n_nodes; id++)
        if (db.tree.nodes[id].parent < 0 && db.tree.root < 0) db.tree.root = (int)id;
    if (db.tree.root < 0) throw std::runtime_error("union: the tree has no root");
    // ---- hash ----
    const P outer = chash->get("hash");
    if (!outer || outer->kind != Node::OBJECT) throw std::runtime_error("union: CustomHash_v4_FastUtil81 without its map");
    const P osize = outer->get("size");
    const int64_t n_keys = osize ? osize->i : -1;
    if (n_keys < 0 || have_key || rows_seen != (uint64_t)n_keys)
        throw std::runtime_error("union: outer map announces " + std::to_string(n_keys) + " entries, stream holds " + std::to_string(rows_seen));
    return db;
}

}  // namespace rkh
