// rk_place -- FASTA queries + a `--jsondb` database dump or the reference's own `.union` file -> .jplace through
// librappas_place.so, no JVM and no Python.
// The reference's `-p p` phase for one query file (src/main_v2/Main_PLACEMENT_v07.java:150-320): ingest and the jplace writer
// are rk_hostio.hpp, the placement itself is rk_place_batch (GPU; there is no CPU fallback).
// Same options and byte-identical output as `python -m rappas_amd.tools.place`.
#include <chrono>
#include <filesystem>
#include <fstream>
#include <iostream>
#include <sstream>

#include "../../../include/rappas_place.h"
#include "rk_hostio.hpp"

static std::string slurp(const std::string &path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("cannot open " + path);
    std::ostringstream ss;
    ss << f.rdbuf();
    return ss.str();
}

static int usage() {
    std::cerr << "usage: rk_place (--jsondb DB.json | --uniondb DB.union) --fasta READS.fa --out OUT.jplace [--keep-at-most 7] [--keep-factor 0.01]\n"
                 "                [--amb mean|max|skip] [--nsbound X] [--guppy-compat] [--device 0] [--logs DIR]\n"
                 "       rk_place --emit-tree TREE.nwk | --format-float X | --format-double X | --dedup READS.fa | --md5 TEXT\n";
    return 2;
}

int main(int argc, char **argv) {
    try {
        std::string jsondb, uniondb, fasta, out, amb = "mean", logs;
        bool logs_given = false;
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
        if ((jsondb.empty() == uniondb.empty()) || fasta.empty() || out.empty()) return usage();
        uint32_t amb_mode;
        if (amb == "mean") amb_mode = RK_AMB_MEAN; else if (amb == "max") amb_mode = RK_AMB_MAX; else if (amb == "skip") amb_mode = RK_AMB_SKIP;
        else return usage();

        rkh::JsonDb jd;
        rkh::UnionDb ud;
        rkh::Tree tree;
        rk_db_desc d;
        memset(&d, 0, sizeof(d));
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
        rk_db *db = nullptr;
        if (rk_db_create(&d, &db) != RK_OK) throw std::runtime_error(std::string("rk_db_create: ") + rk_last_error());

        const std::vector<rkh::Fasta> records = rkh::read_fasta(slurp(fasta));
        const rkh::Dedup dd = rkh::dedup_index(records);
        const std::vector<std::vector<std::string>> names = rkh::dedup_names(records, dd);
        const size_t n = dd.first_rec.size();
        std::string seq;
        std::vector<uint64_t> off(n + 1, 0);
        for (size_t i = 0; i < n; i++) off[i + 1] = off[i] + records[dd.first_rec[i]].seq.size();
        seq.reserve(off[n]);
        for (size_t i = 0; i < n; i++) seq += records[dd.first_rec[i]].seq;
        const uint32_t K = keep_at_most;
        std::vector<uint8_t> n_rows(n);
        std::vector<uint16_t> branch(n * K);
        std::vector<float> score(n * K);
        std::vector<double> lwr(n * K);
        std::vector<uint32_t> flags(n);
        rk_params p{K, keep_factor, amb_mode, nsbound};
        rk_result res{n_rows.data(), branch.data(), score.data(), lwr.data(), flags.data()};
        rk_counters ct;
        const int rc = rk_place_batch(db, &p, n, (const uint8_t *)seq.data(), off.data(), &res, &ct);
        rk_db_destroy(db);
        if (rc != RK_OK) throw std::runtime_error(std::string("rk_place_batch: ") + rk_last_error());

        const auto pl = rkh::jplace_placements(tree, names, n, K, n_rows.data(), branch.data(), score.data(), lwr.data(), guppy);
        std::ofstream of(out, std::ios::binary);
        if (!of) throw std::runtime_error("cannot write " + out);
        of << rkh::jplace_document(tree, pl, call, guppy);
        {   // notplaced_<query>.tsv under logs/ next to the output, as the reference's workdir/logs (Main_PLACEMENT_v07.java:208-214)
            namespace fs = std::filesystem;
            const fs::path dir = logs_given ? fs::path(logs) : fs::absolute(fs::path(out)).parent_path() / "logs";
            fs::create_directories(dir);
            std::ofstream nf(dir / ("notplaced_" + fs::path(fasta).filename().string() + ".tsv"), std::ios::binary);
            if (!nf) throw std::runtime_error("cannot write the notplaced log under " + dir.string());
            nf << rkh::notplaced_log(records, dd, flags.data());
        }
        std::cerr << n << " unique reads, " << pl.size() << " placed -> " << out << "\n";
        return 0;
    } catch (const std::exception &e) {
        std::cerr << "rk_place: " << e.what() << "\n";
        return 1;
    }
}
