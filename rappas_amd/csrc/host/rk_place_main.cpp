// rk_place -- FASTA queries + a database (a `--jsondb` dump, the reference's own `.union` file, or -- round 4 -- the engine's own
// image file, `--dbimage`: mmap + upload, no parse) -> .jplace through librappas_place.so, no JVM and no Python.
// The reference's `-p p` phase for one query file (src/main_v2/Main_PLACEMENT_v07.java:150-320): ingest and the jplace writer
// are rk_hostio.hpp, the placement itself is rk_place_batch (GPU; there is no CPU fallback).
// Same options and byte-identical output as `python -m rappas_amd.tools.place`.
#include <chrono>
#include <filesystem>
#include <fstream>
#include <iostream>
#include <sstream>

#include <sys/resource.h>

#include "../../../include/rappas_place.h"
#include "rk_hostio.hpp"
#include "rk_fastio.hpp"

static std::string slurp(const std::string &path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("cannot open " + path);
    std::ostringstream ss;
    ss << f.rdbuf();
    return ss.str();
}

static int usage() {
    std::cerr << "usage: rk_place (--jsondb DB.json | --uniondb DB.union | --dbimage DB.rkimg) --fasta READS.fa --out OUT.jplace [--keep-at-most 7]\n"
                 "                [--keep-factor 0.01] [--amb mean|max|skip] [--nsbound X] [--guppy-compat] [--device 0] [--logs DIR]\n"
                 "                [--threads N] [--md5-dedup] [--classic-io] [--timing] [--save-dbimage DB.rkimg]\n"
                 "       rk_place (--jsondb DB.json | --uniondb DB.union) --save-dbimage DB.rkimg      (no GPU needed)\n"
                 "       rk_place --emit-tree TREE.nwk | --format-float X | --format-double X | --dedup READS.fa | --md5 TEXT\n";
    return 2;
}

int main(int argc, char **argv) {
    try {
        std::string jsondb, uniondb, dbimage, save_image, fasta, out, amb = "mean", logs;
        bool logs_given = false, md5_dedup = false, classic = false, timing = false;
        unsigned threads = 0;
        uint32_t keep_at_most = 7;
        float keep_factor = 0.01f, nsbound = -INFINITY;
        bool guppy = false;
        int device = 0;
        std::string call;
        for (int i = 1; i < argc; i++) call += std::string(" ") + argv[i];
        for (int i = 1; i < argc; i++) {
            const std::string a = argv[i];
            auto val = [&]() -> std::string { if (i + 1 >= argc) throw std::runtime_error("missing value after " + a); return argv[++i]; };
            if (a == "--jsondb") jsondb = val();
            else if (a == "--uniondb") uniondb = val();
            else if (a == "--dbimage") dbimage = val();
            else if (a == "--save-dbimage") save_image = val();
            else if (a == "--threads") threads = (unsigned)std::stoul(val());
            else if (a == "--md5-dedup") md5_dedup = true;
            else if (a == "--classic-io") classic = true;
            else if (a == "--timing") timing = true;
            else if (a == "--fasta") fasta = val();
            else if (a == "--out") out = val();
            else if (a == "--keep-at-most") keep_at_most = (uint32_t)std::stoul(val());
            else if (a == "--keep-factor") keep_factor = std::stof(val());
            else if (a == "--amb") amb = val();
            else if (a == "--nsbound") nsbound = std::stof(val());
            else if (a == "--guppy-compat") guppy = true;
            else if (a == "--device") device = std::stoi(val());
            else if (a == "--logs") { logs = val(); logs_given = true; }
            // ---- host-side pieces on their own (no device needed): used by the CPU tests ----
            else if (a == "--emit-tree") {
                const rkh::Tree t = rkh::parse_newick(slurp(val()));
                std::cout << rkh::jplace_newick(t) << "\n" << rkh::write_newick(t, false, false, false) << "\n" << (t.rooted() ? "rooted" : "unrooted");
                for (const auto &n : t.nodes) std::cout << "\n" << n.id << "\t" << n.label << "\t" << n.jplace_edge << "\t" << n.parent;
                std::cout << "\n";
                return 0;
            } else if (a == "--format-float") { std::cout << rkh::java_float_to_string(strtof(val().c_str(), nullptr)) << "\n"; return 0; }
            else if (a == "--format-double") { std::cout << rkh::java_double_to_string(strtod(val().c_str(), nullptr)) << "\n"; return 0; }
            else if (a == "--md5") {
                for (uint8_t b : rkh::md5(val())) printf("%02x", b);
                printf("\n");
                return 0;
            } else if (a == "--dedup") {
                std::vector<rkh::Fasta> uniq;
                std::vector<std::vector<std::string>> names;
                rkh::dedup_reads(rkh::read_fasta(slurp(val())), uniq, names);
                for (size_t r = 0; r < uniq.size(); r++) {
                    std::cout << uniq[r].seq;
                    for (const auto &n : names[r]) std::cout << "\t" << n;
                    std::cout << "\n";
                }
                return 0;
            } else if (a == "--fast-rate") {  // FASTA [TREE]: the fast path's host passes on their own, timed (no device needed); with a tree also the writer, on made-up placements
                const std::string path = val();
                const std::string tpath = (i + 1 < argc && argv[i + 1][0] != '-') ? val() : std::string();
                auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
                unsigned hw = std::thread::hardware_concurrency();
                rkh::Team team(threads ? threads : std::max(1u, std::min(hw ? hw : 1u, 32u)));
                rkh::MappedFile fa;
                fa.open_file(path);
                for (int rep = 0; rep < 3; rep++) {
                    const double t0 = now();
                    const rkh::FastaScan sc = rkh::scan_fasta(fa.data, fa.size, team, md5_dedup);
                    const double t1 = now();
                    const rkh::FastDedup dd = rkh::dedup_fast(sc, team);
                    const double t2 = now();
                    rkh::RawArray<char> seq;
                    rkh::RawArray<uint64_t> off;
                    rkh::gather_unique(sc, dd, team, seq, off);
                    const double t3 = now();
                    printf("%zu reads, %zu unique, %u threads: scan %.1f ms (%.2f GB/s), dedup %.1f ms, gather %.1f ms", sc.recs.size(), dd.first_rec.size(), team.size(),
                           (t1 - t0) * 1e3, fa.size / (t1 - t0) / 1e9, (t2 - t1) * 1e3, (t3 - t2) * 1e3);
                    if (!tpath.empty()) {
                        const rkh::Tree t = rkh::parse_newick(slurp(tpath));
                        const size_t n = dd.first_rec.size();
                        const uint32_t K = keep_at_most;
                        std::vector<uint8_t> n_rows(n, 3);
                        std::vector<uint16_t> branch(n * K);
                        std::vector<float> score(n * K);
                        std::vector<double> lwr(n * K);
                        uint64_t seed = 7;
                        for (size_t q = 0; q < n * K; q++) {
                            seed = seed * 6364136223846793005ull + 1442695040888963407ull;
                            branch[q] = (uint16_t)((seed >> 33) % t.nodes.size());
                            score[q] = -(float)((seed >> 20) & 0xFFFFF) / 1713.0f;
                            lwr[q] = (double)(seed >> 11) * 0x1.0p-53;
                        }
                        const double t4 = now();
                        const rkh::FastWriteStats ws = rkh::write_jplace_fast("/dev/shm/rk_fast_rate.jplace", t, sc, dd, K, n_rows.data(), branch.data(), score.data(), lwr.data(), call, guppy, team);
                        const double t5 = now();
                        printf(", write %.1f ms (format %.1f, io %.1f; %.0f MB)", (t5 - t4) * 1e3, ws.format_s * 1e3, ws.io_s * 1e3, ws.bytes / 1e6);
                        (void)unlink("/dev/shm/rk_fast_rate.jplace");
                    }
                    printf("\n");
                }
                return 0;
            } else if (a == "--dedup-fast") {  // the fast path's scan + dedup, printed like --dedup (CPU tests hold the two together)
                const std::string path = val();
                rkh::Team team(threads ? threads : 4);
                rkh::MappedFile fa;
                fa.open_file(path);
                const rkh::FastaScan sc = rkh::scan_fasta(fa.data, fa.size, team, md5_dedup);
                const rkh::FastDedup dd = rkh::dedup_fast(sc, team);
                for (size_t u = 0; u < dd.first_rec.size(); u++) {
                    uint32_t rec = dd.first_rec[u];
                    std::cout << std::string(sc.recs[rec].seq, sc.recs[rec].seq_len);
                    bool first = true;
                    while (rec != 0xFFFFFFFFu) {
                        std::string name(sc.recs[rec].hdr, sc.recs[rec].hdr_len);
                        if (!first) { const size_t cut = name.find(' '); if (cut != std::string::npos) name.resize(cut); }
                        std::cout << "\t" << name;
                        first = false;
                        rec = dd.next_dup[rec];
                    }
                    std::cout << "\n";
                }
                return 0;
            } else if (a == "--write-selftest") {  // FASTA TREE OUT_FAST OUT_CLASSIC SEED: both jplace writers on the same made-up placements
                const std::string fpath = val(), tpath = val(), out_fast = val(), out_classic = val();
                uint64_t seed = std::stoull(val());
                auto rnd = [&]() { seed = seed * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(seed >> 33); };
                const rkh::Tree t = rkh::parse_newick(slurp(tpath));
                rkh::Team team(threads ? threads : 4);
                rkh::MappedFile fa;
                fa.open_file(fpath);
                const rkh::FastaScan sc = rkh::scan_fasta(fa.data, fa.size, team, md5_dedup);
                const rkh::FastDedup dd = rkh::dedup_fast(sc, team);
                const size_t n = dd.first_rec.size();
                const uint32_t K = keep_at_most;
                std::vector<uint8_t> n_rows(n);
                std::vector<uint16_t> branch(n * K);
                std::vector<float> score(n * K);
                std::vector<double> lwr(n * K);
                for (size_t i = 0; i < n; i++) {
                    n_rows[i] = (uint8_t)(rnd() % (K + 1));
                    if (rnd() % 7 == 0) n_rows[i] = 0;
                    for (uint32_t j = 0; j < K; j++) {
                        branch[i * K + j] = (uint16_t)(rnd() % t.nodes.size());
                        score[i * K + j] = -(float)(rnd() % 100000) / 97.0f - (j ? 0.0f : 1e-3f * (float)(rnd() % 10));
                        lwr[i * K + j] = (double)(rnd() % 1000003) / 1000003.0 * (rnd() % 5 == 0 ? 1e-9 : 1.0);
                    }
                }
                const rkh::FastWriteStats ws = rkh::write_jplace_fast(out_fast, t, sc, dd, K, n_rows.data(), branch.data(), score.data(), lwr.data(), call, guppy, team);
                // classic: the records as rk_hostio.hpp sees them
                const std::vector<rkh::Fasta> records = rkh::read_fasta(slurp(fpath));
                const rkh::Dedup cd = rkh::dedup_index(records);
                if (cd.first_rec.size() != n) throw std::runtime_error("the two dedups disagree on the number of unique reads");
                const auto names = rkh::dedup_names(records, cd);
                const auto pl = rkh::jplace_placements(t, names, n, K, n_rows.data(), branch.data(), score.data(), lwr.data(), guppy);
                std::ofstream of(out_classic, std::ios::binary);
                of << rkh::jplace_document(t, pl, call, guppy);
                std::cout << n << " " << ws.placed << " " << (ws.exact_path ? "exact" : "direct") << "\n";
                return 0;
            } else if (a == "--ingest-rate") {  // N3 throughput: FASTA parse, MD5 dedup, host-side packing (no device needed)
                const std::string text = slurp(val());
                auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
                const double t0 = now();
                const std::vector<rkh::Fasta> recs = rkh::read_fasta(text);
                const double t1 = now();
                const rkh::Dedup dd = rkh::dedup_index(recs);
                const double t2 = now();
                std::cout << recs.size() << " reads, " << dd.first_rec.size() << " unique, " << text.size() << " bytes\n"
                          << "parse " << recs.size() / (t1 - t0) / 1e6 << " Mreads/s (" << text.size() / (t1 - t0) / 1e6 << " MB/s)\n"
                          << "dedup " << recs.size() / (t2 - t1) / 1e6 << " Mreads/s\n";
                return 0;
            } else if (a == "--uniondb-stats") {  // load a `.union` file and say what it cost: rows, entries, peak resident memory (CPU tests)
                const std::string path = val();
                size_t rows, entries, nodes;
                {
                    const rkh::UnionDb db = rkh::load_uniondb(slurp(path));
                    rows = db.key_codes.size(); entries = db.scores.size(); nodes = db.tree.nodes.size();
                }
                struct rusage ru;
                getrusage(RUSAGE_SELF, &ru);
                std::cout << rows << " " << entries << " " << nodes << " " << ru.ru_maxrss << "\n";  // (ru_maxrss: kilobytes on Linux)
                return 0;
            } else if (a == "--load-uniondb") {  // what load_uniondb makes of a `.union` stream (compared with the Python twin)
                const rkh::UnionDb db = rkh::load_uniondb(slurp(val()));
                std::cout << db.alphabet << " " << db.k << " " << (db.convert_uo ? 1 : 0) << " " << (db.only_fakes ? 1 : 0) << " "
                          << rkh::java_float_to_string(db.thr) << " " << rkh::java_float_to_string(db.thr_log10) << " "
                          << rkh::java_float_to_string(db.omega) << " " << rkh::java_float_to_string(db.calibration) << " "
                          << db.key_codes.size() << " " << db.scores.size() << "\n" << rkh::jplace_newick(db.tree) << "\n";
                for (const auto &n : db.tree.nodes) std::cout << n.id << "\t" << n.label << "\t" << n.jplace_edge << "\t" << n.parent << "\n";
                for (size_t r = 0; r < db.key_codes.size(); r++) {
                    std::cout << db.key_codes[r];
                    for (uint64_t e = db.row_offsets[r]; e < db.row_offsets[r + 1]; e++)
                        std::cout << " " << db.branch_ids[e] << ":" << rkh::java_float_to_string(db.scores[e]);
                    std::cout << "\n";
                }
                return 0;
            } else if (a == "--load-jsondb") {
                const rkh::JsonDb db = rkh::load_jsondb(slurp(val()));
                std::cout << db.k << " " << rkh::java_float_to_string(db.thr) << " " << rkh::java_float_to_string(db.thr_log10) << " "
                          << db.key_codes.size() << " " << db.scores.size() << "\n" << db.original_tree << "\n";
                for (size_t r = 0; r < db.key_codes.size(); r++) {
                    std::cout << db.key_codes[r];
                    for (uint64_t e = db.row_offsets[r]; e < db.row_offsets[r + 1]; e++)
                        std::cout << " " << db.branch_ids[e] << ":" << rkh::java_float_to_string(db.scores[e]);
                    std::cout << "\n";
                }
                return 0;
            } else return usage();
        }
        const int n_sources = (jsondb.empty() ? 0 : 1) + (uniondb.empty() ? 0 : 1) + (dbimage.empty() ? 0 : 1);
        const bool only_convert = !save_image.empty() && fasta.empty() && out.empty();
        if (n_sources != 1 || (!only_convert && (fasta.empty() || out.empty())) || (only_convert && !dbimage.empty())) return usage();
        uint32_t amb_mode;
        if (amb == "mean") amb_mode = RK_AMB_MEAN; else if (amb == "max") amb_mode = RK_AMB_MAX; else if (amb == "skip") amb_mode = RK_AMB_SKIP;
        else return usage();
        auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        const double t_start = now();

        rkh::JsonDb jd;
        rkh::UnionDb ud;
        rkh::Tree tree;
        rk_db_desc d;
        memset(&d, 0, sizeof(d));
        rk_db *db = nullptr;
        if (!dbimage.empty()) {
            // the engine's own image: the reference tree travels in its user blob (rkh::tree_to_blob)
            uint64_t len = 0;
            if (rk_db_image_user(dbimage.c_str(), nullptr, 0, &len) != RK_OK) throw std::runtime_error(std::string("rk_db_image_user: ") + rk_last_error());
            std::string blob((size_t)len, '\0');
            if (len && rk_db_image_user(dbimage.c_str(), blob.data(), len, &len) != RK_OK) throw std::runtime_error(std::string("rk_db_image_user: ") + rk_last_error());
            tree = rkh::tree_from_blob(blob);
            if (rk_db_load(dbimage.c_str(), device, &db) != RK_OK) throw std::runtime_error(std::string("rk_db_load: ") + rk_last_error());
            rk_db_info info;
            (void)rk_db_get_info(db, &info);
            if (info.n_branches != tree.nodes.size()) throw std::runtime_error("database image: the tree has " + std::to_string(tree.nodes.size()) + " nodes, the database " + std::to_string(info.n_branches) + " branches");
        } else {
            if (!uniondb.empty()) {
                ud = rkh::load_uniondb(slurp(uniondb));
                tree = ud.tree;
                d.alphabet = ud.alphabet == 4 ? RK_ALPHABET_DNA : RK_ALPHABET_AA; d.convert_uo = ud.convert_uo ? 1 : 0; d.k = ud.k;
                d.thr_log10 = ud.thr_log10; d.thr = ud.thr; d.n_keys = ud.key_codes.size();
                d.key_codes = ud.key_codes.data(); d.row_offsets = ud.row_offsets.data(); d.branch_ids = ud.branch_ids.data(); d.scores = ud.scores.data();
            } else {
                jd = rkh::load_jsondb(slurp(jsondb));
                tree = rkh::parse_newick(jd.original_tree);
                d.alphabet = RK_ALPHABET_DNA; d.k = jd.k;
                d.thr_log10 = jd.thr_log10; d.thr = jd.thr; d.n_keys = jd.key_codes.size();
                d.key_codes = jd.key_codes.data(); d.row_offsets = jd.row_offsets.data(); d.branch_ids = jd.branch_ids.data(); d.scores = jd.scores.data();
            }
            d.n_branches = (uint32_t)tree.nodes.size();
            d.device = device; d.table_mode = RK_TABLE_AUTO;
            if (!save_image.empty()) {  // (built on the host: a machine without a GPU can write the image a placement node loads)
                const std::string blob = rkh::tree_to_blob(tree);
                if (rk_db_save_desc(&d, save_image.c_str(), blob.data(), blob.size()) != RK_OK) throw std::runtime_error(std::string("rk_db_save_desc: ") + rk_last_error());
                if (only_convert) { std::cerr << "database image -> " << save_image << "\n"; return 0; }
            }
            if (rk_db_create(&d, &db) != RK_OK) throw std::runtime_error(std::string("rk_db_create: ") + rk_last_error());
        }
        struct DbGuard { rk_db *p; ~DbGuard() { if (p) rk_db_destroy(p); } } db_guard{db};
        const double t_db = now();
        const uint32_t K = keep_at_most;
        rk_params p{K, keep_factor, amb_mode, nsbound};
        namespace fs = std::filesystem;
        const fs::path log_dir = logs_given ? fs::path(logs) : fs::absolute(fs::path(out)).parent_path() / "logs";
        const std::string notplaced_name = "notplaced_" + fs::path(fasta).filename().string() + ".tsv";

        if (!classic) {
            // ---- every host thread on every pass (rk_fastio.hpp) ----
            unsigned hw = std::thread::hardware_concurrency();
            rkh::Team team(threads ? threads : std::max(1u, std::min(hw ? hw : 1u, 32u)));
            // the engine's first batch of a process sets up its workspaces and has its kernels loaded to the device (~100 ms): started
            // here, on a thread of its own, next to the mapping and the scan of the query file (engine start-up, like the database load)
            double warm_s = 0;
            std::thread warm([&]() { const double w0 = now(); (void)rk_reserve_host_path(db, K, 320); warm_s = now() - w0; });
            struct JoinWarm { std::thread &t; ~JoinWarm() { if (t.joinable()) t.join(); } } join_warm{warm};
            rkh::MappedFile fa;
            fa.open_file(fasta);
            const double t0 = now();
            const rkh::FastaScan sc = rkh::scan_fasta(fa.data, fa.size, team, md5_dedup);
            const double t1 = now();
            const rkh::FastDedup dd = rkh::dedup_fast(sc, team);
            const double t2 = now();
            rkh::RawArray<char> seq;
            rkh::RawArray<uint64_t> off;
            rkh::gather_unique(sc, dd, team, seq, off);
            const size_t n = dd.first_rec.size();
            rkh::RawArray<uint8_t> n_rows;
            rkh::RawArray<uint16_t> branch;
            rkh::RawArray<float> score;
            rkh::RawArray<double> lwr;
            rkh::RawArray<uint32_t> flags;
            n_rows.alloc(n); branch.alloc(n * K); score.alloc(n * K); lwr.alloc(n * K); flags.alloc(n);
            team.run([&](unsigned t, unsigned T) {  // (first touch by every thread at once, not by the engine's few result-copy threads)
                const size_t lo = n * t / T, hi = n * (t + 1) / T;
                memset(n_rows.data() + lo, 0, hi - lo); memset(flags.data() + lo, 0, (hi - lo) * 4);
                memset(branch.data() + lo * K, 0, (hi - lo) * K * 2); memset(score.data() + lo * K, 0, (hi - lo) * K * 4); memset(lwr.data() + lo * K, 0, (hi - lo) * K * 8);
            });
            const double t3 = now();
            warm.join();
            const double t3b = now();
            rk_result res{n_rows.data(), branch.data(), score.data(), lwr.data(), flags.data()};
            rk_counters ct;
            if (rk_place_batch(db, &p, n, (const uint8_t *)seq.data(), off.data(), &res, &ct) != RK_OK) throw std::runtime_error(std::string("rk_place_batch: ") + rk_last_error());
            const double t4 = now();
            const rkh::FastWriteStats ws = rkh::write_jplace_fast(out, tree, sc, dd, K, n_rows.data(), branch.data(), score.data(), lwr.data(), call, guppy, team);
            const double t5 = now();
            fs::create_directories(log_dir);
            {
                std::ofstream nf(log_dir / notplaced_name, std::ios::binary);
                if (!nf) throw std::runtime_error("cannot write the notplaced log under " + log_dir.string());
                nf << rkh::notplaced_log_fast(sc, dd, flags.data());
            }
            const double t6 = now();
            std::cerr << n << " unique reads, " << ws.placed << " placed -> " << out << "\n";
            if (timing) {  // one JSON line (bench.py's fasta_to_jplace leg reads it): seconds per pass, FASTA bytes in -> jplace bytes out
                char buf[700];
                snprintf(buf, sizeof(buf),
                         "{\"reads\": %zu, \"unique\": %zu, \"placed\": %llu, \"fasta_bytes\": %zu, \"jplace_bytes\": %llu, \"threads\": %u, \"db_s\": %.6f, \"engine_warm_up_s\": %.6f, \"scan_s\": %.6f, "
                         "\"dedup_s\": %.6f, \"gather_s\": %.6f, \"place_s\": %.6f, \"place_wait_for_warm_up_s\": %.6f, \"write_s\": %.6f, \"write_format_s\": %.6f, \"write_io_s\": %.6f, \"notplaced_log_s\": %.6f, \"fasta_to_jplace_s\": %.6f, "
                         "\"exact_writer\": %s}",
                         sc.recs.size(), n, (unsigned long long)ws.placed, fa.size, (unsigned long long)ws.bytes, team.size(), t_db - t_start, warm_s, t1 - t0, t2 - t1, t3 - t2,
                         t4 - t3, t3b - t3, t5 - t4, ws.format_s, ws.io_s, t6 - t5, t5 - t0, ws.exact_path ? "true" : "false");
                std::cout << buf << std::endl;
            }
            return 0;
        }

        // ---- the one-string-at-a-time path of rk_hostio.hpp (the definition the fast path is held to) ----
        const double tc0 = now();
        const std::vector<rkh::Fasta> records = rkh::read_fasta(slurp(fasta));
        const rkh::Dedup dd = rkh::dedup_index(records);
        const std::vector<std::vector<std::string>> names = rkh::dedup_names(records, dd);
        const size_t n = dd.first_rec.size();
        std::string seq;
        std::vector<uint64_t> off(n + 1, 0);
        for (size_t i = 0; i < n; i++) off[i + 1] = off[i] + records[dd.first_rec[i]].seq.size();
        seq.reserve(off[n]);
        for (size_t i = 0; i < n; i++) seq += records[dd.first_rec[i]].seq;
        std::vector<uint8_t> n_rows(n);
        std::vector<uint16_t> branch(n * K);
        std::vector<float> score(n * K);
        std::vector<double> lwr(n * K);
        std::vector<uint32_t> flags(n);
        rk_result res{n_rows.data(), branch.data(), score.data(), lwr.data(), flags.data()};
        rk_counters ct;
        const int rc = rk_place_batch(db, &p, n, (const uint8_t *)seq.data(), off.data(), &res, &ct);
        if (rc != RK_OK) throw std::runtime_error(std::string("rk_place_batch: ") + rk_last_error());

        const auto pl = rkh::jplace_placements(tree, names, n, K, n_rows.data(), branch.data(), score.data(), lwr.data(), guppy);
        std::ofstream of(out, std::ios::binary);
        if (!of) throw std::runtime_error("cannot write " + out);
        of << rkh::jplace_document(tree, pl, call, guppy);
        {   // notplaced_<query>.tsv under logs/ next to the output, as the reference's workdir/logs (Main_PLACEMENT_v07.java:208-214)
            fs::create_directories(log_dir);
            std::ofstream nf(log_dir / notplaced_name, std::ios::binary);
            if (!nf) throw std::runtime_error("cannot write the notplaced log under " + log_dir.string());
            nf << rkh::notplaced_log(records, dd, flags.data());
        }
        std::cerr << n << " unique reads, " << pl.size() << " placed -> " << out << "\n";
        if (timing) std::cout << "{\"reads\": " << records.size() << ", \"unique\": " << n << ", \"db_s\": " << (t_db - t_start) << ", \"fasta_to_jplace_s\": " << (now() - tc0) << ", \"classic\": true}" << std::endl;
        return 0;
    } catch (const std::exception &e) {
        std::cerr << "rk_place: " << e.what() << "\n";
        return 1;
    }
}
