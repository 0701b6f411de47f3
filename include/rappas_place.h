/*
 * rappas_place.h -- C ABI of the MI355X-native phylo-kmer placement engine.
 *
 * Drop-in boundary for RAPPAS's query-placement hot path (`-p p`).  The reference (Java) has no
 * FFI seam; the boundary is cut where SURVEY.md section 8(b) puts it.  Paths below are relative to
 * the reference root.  Each entry point names the reference code it replaces; INTEGRATION.md shows
 * the JNI stub a RAPPAS maintainer would add on the Java side.
 *
 *   rk_db_create          <- the lookup structure behind session.hash:
 *                            src/core/hash/CustomHash_v4_FastUtil81.java:36,146-153 (+ HashStrategy.java:22-29)
 *                            fed from the walk SessionNext_v2.saveToJSON does (src/main_v2/SessionNext_v2.java:250-261)
 *                            and the scalars of src/main_v2/SessionNext_v2.java:43-66 (k, states, PPStarThreshold*).
 *   rk_place_batch        <- the per-read body of PlacementProcess.processQueries,
 *                            src/core/algos/PlacementProcess.java:645-1025 (knife init, k-mer loop, accumulate,
 *                            ambiguity mean/max :1129-1236, fillBestScoreList :396-451, LWR + keep-factor :974-1025),
 *                            with src/core/algos/AmbigSequenceKnife.java:98-272 and
 *                            src/core/DNAStatesShifted.java:115-143,182-243 / src/core/AAStates.java:48-197 underneath.
 *   rk_pack_reads_device  <- AmbigSequenceKnife.initTables char->state part (AmbigSequenceKnife.java:103-130) as a
 *                            device kernel: ASCII -> 2-bit / 5-bit packed records + per-read flags.
 *   rk_place_packed_device<- same body as rk_place_batch for reads already packed and resident in HBM.
 *
 * Conventions: plain C types only; inputs are borrowed for the duration of a call, outputs are
 * caller-owned; return 0 on success, <0 on error with a message in rk_last_error() (thread-local);
 * the library never calls exit().  The product path has no CPU fallback: every entry point that
 * computes placements requires a HIP device and fails with RK_ERR_NO_DEVICE otherwise.
 */
#ifndef RAPPAS_PLACE_H
#define RAPPAS_PLACE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RK_VERSION 101 /* 0.1.1: rk_db_save / rk_db_save_desc / rk_db_load / rk_db_image_info / rk_db_image_user, rk_reserve_host_path, rk_count_work_device, RK_ERR_IO */

/* alphabets = number of unambiguous states (States.getNonAmbiguousStatesCount()) */
#define RK_ALPHABET_DNA 4  /* src/core/DNAStatesShifted.java : A=0 T/U=1 C=2 G=3, 2 bits/base   */
#define RK_ALPHABET_AA 20  /* src/core/AAStates.java : R H K D E S T N Q C G P A I L M F W Y V, 5 bits/residue */

/* ambiguity handling (src/main_v2/ArgumentsParser_v2.java:90-91; PlacementProcess.java:738-749) */
#define RK_AMB_SKIP 0 /* --noamb      */
#define RK_AMB_MEAN 1 /* default      */
#define RK_AMB_MAX 2  /* --ambwithmax */

/* k-mer -> row lookup structure */
#define RK_TABLE_AUTO 0   /* direct when sigma^k slots fit the budget, else hash */
#define RK_TABLE_HASH 1   /* open-addressed (linear probing), 16-byte slots {key, row descriptor} */
#define RK_TABLE_DIRECT 2 /* identity-hashed, collision-free special case over all sigma^k codes: compact 16-byte blocks of
                             12 k-mers (1.33 bytes per k-mer, L2-resident); falls back to DIRECT8 if a row exceeds 2040 entries */
#define RK_TABLE_DIRECT8 4 /* identity-hashed, one 8-byte row descriptor per code */

/* per-read result flags */
#define RK_FLAG_PLACED 1u         /* >=1 k-mer matched (L non-empty, PlacementProcess.java:797) */
#define RK_FLAG_BAD_CHAR 2u       /* unsupported character; the reference would System.exit(1) (AmbigSequenceKnife.java:124-128) */
#define RK_FLAG_TOO_SHORT 4u      /* R < k : no k-mer (the reference crashes for R < k-1) */
#define RK_FLAG_AMBIGUOUS 8u      /* read contains an ambiguity character */
#define RK_FLAG_BELOW_NSBOUND 16u /* best score < ns_bound: no jplace record (PlacementProcess.java:974) */
#define RK_FLAG_TOO_LONG 64u      /* device pack only: read longer than the packed record; not placed */

/* error codes */
#define RK_OK 0
#define RK_ERR_INVALID -1
#define RK_ERR_NO_DEVICE -2
#define RK_ERR_HIP -3
#define RK_ERR_NOMEM -4
#define RK_ERR_UNSUPPORTED -5
#define RK_ERR_IO -6 /* a database image file could not be read / written, or failed its size / checksum tests */

typedef struct rk_db rk_db;

typedef struct rk_db_desc {
    uint32_t alphabet;    /* RK_ALPHABET_DNA | RK_ALPHABET_AA */
    uint32_t convert_uo;  /* AA only: U->C, O->L (src/core/AAStates.java:118-123) */
    uint32_t k;           /* DNA: 2..31 (hashed table from k = 16: 4^k codes), AA: 2..12 */
    uint32_t n_branches;  /* originalTree.getNodeCount() (PlacementProcess.java:495-496); branch ids < n_branches <= 65535 */
    float thr_log10;      /* session.PPStarThresholdAsLog10 (T) */
    float thr;            /* session.PPStarThreshold (P), used by the ambiguity-mean path */
    uint64_t n_keys;
    const uint64_t *key_codes;   /* [n_keys] DNA: sum state_i<<(2i) (== little-endian compressMer bytes); AA: sum state_i<<(5i) */
    const uint64_t *row_offsets; /* [n_keys+1] CSR offsets into branch_ids/scores */
    const uint16_t *branch_ids;  /* [n_entries] (char)nodeId of CustomHash_v4_FastUtil81.java:79,87; unique within a row */
    const float *scores;         /* [n_entries] log10 PP*, finite */
    int32_t device;              /* HIP device ordinal */
    uint32_t table_mode;         /* RK_TABLE_* */
} rk_db_desc;

typedef struct rk_db_info {
    uint32_t alphabet, k, n_branches, table_mode;
    float thr_log10, thr;
    uint64_t n_keys, n_entries;
    uint64_t table_slots, table_bytes, rows_bytes; /* HBM footprint */
    uint32_t bits_per_symbol, max_row_len;
    int32_t device;
} rk_db_info;

typedef struct rk_params {
    uint32_t keep_at_most; /* --keep-at-most, default 7 (ArgumentsParser_v2.java:87); 1..16 */
    float keep_factor;     /* --keep-factor, default 0.01f (ArgumentsParser_v2.java:88) */
    uint32_t amb_mode;     /* RK_AMB_* */
    float ns_bound;        /* --nsbound / calibrationNormScore, default -INFINITY */
} rk_params;

typedef struct rk_counters {
    uint64_t reads, placed, unplaced, bad_char, too_short, ambiguous;
} rk_counters;

/* Result arrays, n_reads x keep_at_most, rows ordered best -> worse and already cut by keep_factor.
 * Unused rows: branch 0xFFFF, score -inf, lwr 0.  Host pointers for rk_place_batch, device pointers
 * for rk_place_packed_device.  Tie rule (the reference's is map-layout dependent): score desc, branch id asc. */
typedef struct rk_result {
    uint8_t *n_rows;   /* [n_reads]   rows emitted (0 => unplaced or gated) */
    uint16_t *branch;  /* [n_reads*K] original-tree node id of the edge's child */
    float *score;      /* [n_reads*K] S[x], bit-exact float32 */
    double *lwr;       /* [n_reads*K] likelihood weight ratio */
    uint32_t *flags;   /* [n_reads]   RK_FLAG_* */
} rk_result;

int rk_version(void);
const char *rk_last_error(void);

/* Main_DBBUILD_3.java:165-166 (float32 threshold pair from omega, #states, k) */
void rk_thresholds(float omega, uint32_t n_states, uint32_t k, float *thr, float *thr_log10);

int rk_db_create(const rk_db_desc *desc, rk_db **out);
/* Same argument checks and host-side image construction as rk_db_create, but no device is touched: lets the caller
 * (or a CPU-only test) validate a DB and learn its HBM footprint / table flavour.  info may be NULL. */
int rk_db_validate(const rk_db_desc *desc, rk_db_info *info);
void rk_db_destroy(rk_db *db);
/* Another handle of the same database on `device` (it may be the source's own device), copied device to device -- over xGMI
 * between GPUs -- instead of being rebuilt and uploaded once per GPU: what a single-process caller (one JVM, rk_place_batch_multi)
 * does after the first rk_db_create, and the only way to replicate an image that exists in HBM only (rk_db_create_synth). */
int rk_db_clone(const rk_db *src, int32_t device, rk_db **out);
int rk_db_get_info(const rk_db *db, rk_db_info *info);

/* The database as a file: the HBM image exactly as the kernels read it (k-mer table, row blob, window spans) behind a fixed header,
 * so that loading is mmap + one host-to-device copy per section -- no parse, no rebuild.  Stands where the reference stores and
 * reloads its Java-serialised session (src/main_v2/SessionNext_v2.java:110-154 storeHash, :158-207 load); only the lookup structure
 * and the scalars of rk_db_desc are kept -- whatever else the caller needs next to it (rk_place: the reference tree) travels as an
 * opaque `user` blob the engine never looks inside.
 *   rk_db_save        a handle's image, read back from the device.
 *   rk_db_save_desc   the same file from the caller's CSR arrays, built on the host: no device is touched (desc->device is ignored).
 *   rk_db_load        a new handle on `device`; size, header and payload checksums are verified before the device is looked at, a
 *                     truncated / overwritten / bit-flipped file is refused with RK_ERR_IO.  (Integrity, not authenticity: an image
 *                     is trusted input, like the library itself.)
 *   rk_db_image_info  the same checks without a device; info (device = -1) and the user blob's length, either may be NULL.
 *   rk_db_image_user  the user blob: at most `cap` bytes into buf, *len = its full length.
 * Files are little-endian and specific to the image version this library writes (a newer / older file is refused, not guessed at). */
int rk_db_save(const rk_db *db, const char *path, const void *user, uint64_t user_bytes);
int rk_db_save_desc(const rk_db_desc *desc, const char *path, const void *user, uint64_t user_bytes);
int rk_db_load(const char *path, int32_t device, rk_db **out);
int rk_db_image_info(const char *path, rk_db_info *info, uint64_t *user_bytes);
int rk_db_image_user(const char *path, void *buf, uint64_t cap, uint64_t *len);

/* One row read back out of the HBM image through the same table lookup and entry decode the placement kernels use:
 * the engine's counterpart of CustomHash_v4_FastUtil81.getPairsOfTopPosition2 (src/core/hash/CustomHash_v4_FastUtil81.java:146-153;
 * null there <=> *len == 0 here).  Entries come back in the image's order (large-tree images: ascending branch id).  At most
 * `cap` entries are written; *len is the row's full length.  For checkers and tools, not for the hot path (one launch per call). */
int rk_db_fetch_row(rk_db *db, uint64_t code, uint32_t cap, uint32_t *len, uint16_t *branch_ids, float *scores);

/* The seeded synthetic database of the measurement plan (keys = a random subset of the code space,
 * row length 1 + geometric, branch ids a contiguous window, scores v = T*u), generated ON THE DEVICE straight into the HBM
 * image -- the only way a C5-class (~200 GB) database can exist, since no host holds its CSR form.  Every value is a pure
 * function of (seed, dense k-mer index, entry index) in integer arithmetic plus one float32 multiply (definition at the top of
 * rappas_amd/csrc/rk_synth_impl.h), so a checker regenerates any row on the host (rappas_amd/synth.py: synth_rows) and
 * rk_db_fetch_row reads it back.  The reference has no counterpart (its databases come out of `-p b`); this is bench / test
 * input, built by the same library because only the library knows the image format. */
typedef struct rk_synth_desc {
    uint32_t alphabet, convert_uo, k, n_branches;
    float thr_log10, thr;
    uint64_t seed;
    double key_fraction;  /* probability that a code of the k-mer space carries a row, in (0, 1] */
    double mean_row_len;  /* rows are 1 + geometric with this mean, capped at n_branches - 1 */
    int32_t device;
    uint32_t table_mode;  /* RK_TABLE_* */
} rk_synth_desc;
int rk_db_create_synth(const rk_synth_desc *desc, rk_db **out);

/* Host-buffer entry point: ASCII reads (concatenated, seq_off[n_reads+1]) -> results in host memory.
 * Internally: H2D, device-side pack, placement kernel(s), D2H; chunked to bound device memory. */
int rk_place_batch(rk_db *db, const rk_params *p, uint64_t n_reads, const uint8_t *seq_ascii,
                   const uint64_t *seq_off, rk_result *out, rk_counters *counters);

/* Optional: sets up ahead of time what the first rk_place_batch / rk_place_batch_packed of a handle otherwise sets up on its way
 * (streams, device buffers and page-locked staging for full chunks of reads of up to max_read_len symbols: ~80 ms), e.g. while the
 * caller is still reading its input.  Nothing is placed. */
int rk_reserve_host_path(rk_db *db, uint32_t keep_at_most, uint32_t max_read_len);

/* The same for reads the host has already packed (2 bits per base / 5 per residue, symbol i at bits [i*b, (i+1)*b) of the
 * record's little-endian bit string -- the layout rk_pack_reads_device produces): 38 instead of 150 bytes per 150-bp read cross
 * PCIe.  lens NULL => every read has fixed_len symbols; flags NULL => no read carries BAD_CHAR / AMBIGUOUS.  seq_ascii / seq_off
 * (both or neither) are consulted only for chunks that hold a read flagged AMBIGUOUS (the ambiguity path of
 * PlacementProcess.java:1129-1236 works on characters); without them such reads come back unplaced with the flag set.
 * rk_pack_reads_host is the matching host-side packer (AmbigSequenceKnife.java:103-130 char -> state, threaded; n_threads 0 =
 * auto): it writes the records, lengths and flags exactly as the device packer does. */
int rk_place_batch_packed(rk_db *db, const rk_params *p, uint64_t n_reads, const uint32_t *packed, uint32_t words_per_read,
                          const uint32_t *lens, uint32_t fixed_len, const uint32_t *flags, const uint8_t *seq_ascii,
                          const uint64_t *seq_off, rk_result *out, rk_counters *counters);
int rk_pack_reads_host(const rk_db *db, uint64_t n_reads, const uint8_t *seq_ascii, const uint64_t *seq_off, uint32_t words_per_read,
                       uint32_t *packed, uint32_t *lens, uint32_t *flags, uint32_t n_threads);
/* The same packer without a database handle (no GPU involved): `alphabet` RK_ALPHABET_DNA / RK_ALPHABET_AA, `convert_uo` the
 * database's --convertUO switch (AAStates.java:118-123), `k` for RK_FLAG_TOO_SHORT.  words_per_read >= ceil(longest read * b / 32),
 * b = 2 (DNA) / 5 (amino acids); longer reads are cut and flagged RK_FLAG_TOO_LONG. */
int rk_pack_reads(uint32_t alphabet, int convert_uo, uint32_t k, uint64_t n_reads, const uint8_t *seq_ascii, const uint64_t *seq_off,
                  uint32_t words_per_read, uint32_t *packed, uint32_t *lens, uint32_t *flags, uint32_t n_threads);

/* The same over several GPUs from ONE host process (RAPPAS is a single JVM): dbs[g] are handles of the same database created
 * on different devices (rk_db_create with desc.device = g); the batch is cut into n_dbs contiguous shards, shard g goes to
 * dbs[g] on its own host thread, and every shard writes its slice of the caller's result arrays -- reads are independent
 * (PlacementProcess.java:1067-1075 resets all per-read state), so there is no exchange step and no collective.  Results are
 * identical to one rk_place_batch call over the whole batch.  If a shard's device fails (a HIP error, out of memory) the shard is
 * placed again on the handles that finished, in a fresh host thread; the call then still returns RK_OK and rk_last_error()
 * names the device that dropped out ("" when nothing failed).  The call fails only when no healthy device is left for a shard. */
int rk_place_batch_multi(rk_db *const *dbs, uint32_t n_dbs, const rk_params *p, uint64_t n_reads, const uint8_t *seq_ascii,
                         const uint64_t *seq_off, rk_result *out, rk_counters *counters);

/* Page-locked host memory for the buffers handed to rk_place_batch: the DMA then reads / writes them directly.  Pageable
 * buffers work too (they are staged through page-locked memory inside the library, ~1.2e8 reads/s either way on C2); pinned
 * ones save the host threads that staging keeps busy.  A JVM can wrap the allocation with NewDirectByteBuffer. */
void *rk_host_alloc(uint64_t bytes);
void rk_host_free(void *p);

/* Packed-record geometry for a given maximum read length: 32-bit words per record. */
uint32_t rk_packed_words(const rk_db *db, uint32_t max_len);

/* Device entry points (all pointers are device pointers on db's device; stream = hipStream_t or NULL).
 * rk_pack_reads_device: ASCII -> packed records [n_reads][words_per_read] + lens[n_reads] + flags[n_reads]. */
int rk_pack_reads_device(rk_db *db, uint64_t n_reads, const uint8_t *d_seq_ascii, const uint64_t *d_seq_off,
                         uint32_t words_per_read, uint32_t *d_packed, uint32_t *d_lens, uint32_t *d_flags,
                         void *stream);
/* rk_place_packed_device: d_lens may be NULL (every read has fixed_len symbols); d_flags_in may be NULL
 * (no read carries BAD_CHAR/AMBIGUOUS).  Reads flagged AMBIGUOUS need d_seq_ascii/d_seq_off (else they are
 * reported unplaced with the flag set).  Asynchronous on `stream`.  d_flags_in may be the output flag array itself (in place).
 * What the call allocates: nothing from the device's memory pools; the handle keeps one grow-only scratch block per stream it has
 * been launched on (the order the kernels take a large batch's reads in, the marks of tiles one kernel hands to the next: ~5 bytes a
 * read, plain hipMalloc, freed by rk_db_destroy) -- so calls on ONE stream must not overlap in time from different threads, and a
 * call made while the stream is being captured into a graph does without the block if it would have to grow. */
int rk_place_packed_device(rk_db *db, const rk_params *p, uint64_t n_reads, const uint32_t *d_packed,
                           uint32_t words_per_read, const uint32_t *d_lens, uint32_t fixed_len,
                           const uint32_t *d_flags_in, const uint8_t *d_seq_ascii, const uint64_t *d_seq_off,
                           const rk_result *d_out, void *stream);

/* Optional diagnostics (round 4): the work a batch of packed reads asks of the database, counted by a kernel of its own -- the
 * placement kernels carry no counters.  kmers_probed = sum of sk.getMerCount() (AmbigSequenceKnife.java:191) over the reads the
 * packed kernels place (not BAD_CHAR / TOO_LONG / AMBIGUOUS, at least k symbols); kmers_hit = those with a row in the database
 * (hash.getPairsOfTopPosition2(word) != null, PlacementProcess.java:705-707); entries = (branch, score) pairs of those rows, i.e.
 * iterations of the loop at PlacementProcess.java:719-735.  d_out is device memory (zeroed by the call on `stream`, then filled). */
typedef struct rk_work {
    uint64_t kmers_probed, kmers_hit, entries;
} rk_work;
int rk_count_work_device(rk_db *db, uint64_t n_reads, const uint32_t *d_packed, uint32_t words_per_read, const uint32_t *d_lens,
                         uint32_t fixed_len, const uint32_t *d_flags_in, rk_work *d_out, void *stream);

/* Launch geometry knob (0 = auto): lanes cooperating on one read (8,16,32,64). For tuning/benchmarks. */
int rk_set_lanes_per_read(rk_db *db, uint32_t lanes);
/* Name of the placement kernel variant the next launch will use (for profiles). */
const char *rk_kernel_name(const rk_db *db);

/* ------------------------------------------------------------------------------------------------------------------
 * Phylo-kmer database construction (`-p b` hot loop).
 * Replaces, for one reference tree, the triple loop of src/main_v2/Main_DBBUILD_3.java:648-750 -- for every tested node,
 * for every alignment position pos in [0, L-k+2), a fresh src/core/algos/WordExplorer_v3.java explorer (:98-199: the
 * branch-and-bound recursion with its running float32 sum) started from every state rank -- together with the insertions
 * of src/core/hash/CustomHash_v4_FastUtil81.java:73-89 (addTuple: per (k-mer, original branch) keep the largest PP*).
 * Input is what src/core/PProbasSorted.java:19-25 holds after the ancestral reconstruction has been parsed: per node and
 * site the states ranked by descending posterior, with log10 posteriors.  Output is the CSR form rk_db_create takes
 * (keys ascending, branches ascending inside a row), in host memory owned by the library.  The reference's
 * registered scores depend on the order of exploration (a running float is incremented and decremented); the kernel
 * replays each explorer's statements in order, one lane per (node, pos), so the scores are bit-identical.
 * ------------------------------------------------------------------------------------------------------------------ */
typedef struct rk_build_desc {
    uint32_t alphabet;            /* RK_ALPHABET_DNA | RK_ALPHABET_AA (word -> key code as for rk_db_desc.key_codes) */
    uint32_t k;                   /* DNA: 2..15, AA: 2..9 (code << 16 | branch must fit 64 bits) */
    uint32_t n_nodes;             /* tested nodes = rows of the posterior table (Main_DBBUILD_3.java:648 nodesTested) */
    uint32_t n_sites;             /* alignment length == PProbasSorted.getSiteCount() */
    uint32_t n_states;            /* PProbasSorted.getStateCount() */
    uint32_t do_gap_jumps;        /* gapJumpsActivated (Main_DBBUILD_3.java:239-258) */
    uint32_t limit_to_1_jump;     /* ArgumentsParser_v2.java:76 (default true) */
    float thr_log10;              /* session.PPStarThresholdAsLog10 */
    const uint8_t *states;        /* [n_nodes][n_sites][n_states] PProbasSorted.states */
    const float *pp_log10;        /* [n_nodes][n_sites][n_states] PProbasSorted.pp, descending along the last axis */
    const uint16_t *node_branch;  /* [n_nodes] original branch id of each tested node (WordExplorer_v3.java:93-94) */
    const uint32_t *gap_off;      /* [n_sites+1] CSR over sites of Alignment.getGapIntervals(); NULL unless do_gap_jumps */
    const int32_t *gap_len;       /* interval lengths */
    int32_t device;               /* HIP device ordinal */
    uint32_t reserved;
} rk_build_desc;

typedef struct rk_built_db {
    uint64_t n_keys, n_entries;
    uint64_t *key_codes;          /* [n_keys] ascending */
    uint64_t *row_offsets;        /* [n_keys+1] */
    uint16_t *branch_ids;         /* [n_entries] ascending inside a row */
    float *scores;                /* [n_entries] */
    uint64_t tuples;              /* addTuple calls ("Tuples explored", Main_DBBUILD_3.java:760) */
    uint64_t visits;              /* exploreWords calls that passed the alignment-limit test */
    double explore_ms, reduce_ms; /* device time of the two stages (HIP events) */
} rk_built_db;

int rk_build_db(const rk_build_desc *desc, rk_built_db *out);
void rk_built_free(rk_built_db *b);

#ifdef __cplusplus
}
#endif
#endif
