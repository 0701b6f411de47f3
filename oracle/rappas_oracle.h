/*
 * rappas_oracle.h -- CPU restatement of RAPPAS's query-placement hot path (`-p p`).
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may build, link, import or call anything in oracle/.
 * The product path (rappas_amd/, include/) never routes through it.
 *
 * PARITY UNPINNED: the reference (Java, /root/reference) ships no tests, fixtures or golden
 * vectors for this path, and it cannot be built or run here (no JVM, fastutil-8.2.2.jar absent).
 * The restatement below is therefore pinned only by hand-derived known-answer vectors
 * (tests/golden/, each derived from the cited reference lines), not by reference outputs.
 *
 * Every function cites the reference file:line it follows (paths relative to the reference root).
 */
#ifndef RAPPAS_ORACLE_H
#define RAPPAS_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RO_ALPHABET_DNA 4
#define RO_ALPHABET_AA 20

/* ambiguity handling: PlacementProcess.java:738-749 (treatAmbiguities / treatAmbiguitiesWithMax) */
#define RO_AMB_SKIP 0 /* --noamb */
#define RO_AMB_MEAN 1 /* default */
#define RO_AMB_MAX 2  /* --ambwithmax */

/* per-read flags (same bit meaning as include/rappas_place.h RK_FLAG_*) */
#define RO_FLAG_PLACED 1u        /* L non-empty: PlacementProcess.java:797 */
#define RO_FLAG_BAD_CHAR 2u      /* AmbigSequenceKnife.java:124-128 would System.exit(1) */
#define RO_FLAG_TOO_SHORT 4u     /* R < k: zero k-mers (R<k-1 crashes the reference) */
#define RO_FLAG_AMBIGUOUS 8u     /* read contains >=1 ambiguity character */
#define RO_FLAG_BELOW_NSBOUND 16u /* PlacementProcess.java:974 gate failed */
#define RO_FLAG_TIE 32u          /* oracle only: exact float tie inside the top-(K+1) touched scores */

typedef struct ro_db ro_db;

/* Main_DBBUILD_3.java:165-166 : PPStarThreshold and its log10, float32 */
void ro_thresholds(float omega, int n_states, int k, float *thr, float *thr_log10);

/* AmbigSequenceKnife.java:95 */
int ro_max_ambig_per_mer(int k, int n_states);

/* DNAStatesShifted.java:182-209 / AAStates.java:48-123.  Returns state (0..sigma-1),
 * 0x80|class for an ambiguity character, 0xFF for an unsupported character. */
uint8_t ro_char_code(int alphabet, int convert_uo, uint8_t c);
/* alternatives of an ambiguity class: DNAStatesShifted.java:45-96, AAStates.java:97-112 */
int ro_amb_alternatives(int alphabet, uint8_t amb_class, uint8_t *out /* >=20 */);

/* DNAStatesShifted.java:115-143 (byte-wise restatement). Returns byte count. */
int ro_compress_mer_dna(const uint8_t *states, int k, uint8_t *out);
/* integer key code used on both sides of the C ABI:
 * DNA: little-endian integer of compressMer bytes == sum state_i << (2 i)
 * AA : sum state_i << (5 i)   (compressMer is the identity, AAStates.java:195-197) */
uint64_t ro_kmer_code(int alphabet, const uint8_t *states, int k);

/* DB = CustomHash_v4_FastUtil81 (hash/CustomHash_v4_FastUtil81.java:36) restated as
 * code -> CSR row; row iteration order = CSR order given here. Copies its inputs. */
ro_db *ro_db_create(int alphabet, int convert_uo, int k, int n_branches, float thr_log10, float thr,
                    uint64_t n_keys, const uint64_t *key_codes, const uint64_t *row_offsets,
                    const uint16_t *branch_ids, const float *scores);
void ro_db_destroy(ro_db *db);

typedef struct {
    uint64_t reads, placed, unplaced, kmers, kmers_hit, entries, amb_kmers, skipped_kmers;
} ro_counters;

/* PlacementProcess.processQueries per-read body (PlacementProcess.java:645-1075), batch form.
 * Outputs are n_reads x keep_at_most, rows best -> worse, truncated by keep_factor.
 * entries_per_read (optional) = H, the number of DB row entries gathered for the read. */
int ro_place_batch(const ro_db *db, int keep_at_most, float keep_factor, int amb_mode, float ns_bound,
                   uint64_t n_reads, const uint8_t *seq, const uint64_t *seq_off, uint8_t *n_rows,
                   uint16_t *branch, float *score, double *lwr, uint32_t *flags,
                   uint32_t *entries_per_read, ro_counters *counters);

/* Debug view of one read: full S vector (NaN for untouched) + L in insertion order. */
int ro_score_vector(const ro_db *db, int amb_mode, const uint8_t *seq, uint64_t len, float *S_out,
                    int32_t *L_out, int32_t *L_size);

#ifdef __cplusplus
}
#endif
#endif
