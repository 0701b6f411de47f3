/*
 * rappas_build_oracle.h -- CPU restatement of RAPPAS's phylo-kmer DB construction loop (`-p b`), SURVEY.md section 8(f) row N4.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE (same rules as rappas_oracle.h).
 * PARITY UNPINNED: no reference outputs exist for this path either; pinned by hand-derived known answers and by an
 * independent Python restatement (tests/pyref_build.py).
 *
 * What it restates (paths relative to the reference root):
 *   src/main_v2/Main_DBBUILD_3.java:648-750   for every tested node, for pos in [0, L-k+2): a fresh WordExplorer_v3,
 *                                              exploreWords(pos, j) for every state index j
 *   src/core/algos/WordExplorer_v3.java:98-199 the branch-and-bound recursion, INCLUDING its float bookkeeping: one running
 *                                              float `currentLogSum` that is incremented on the way down and decremented on
 *                                              the way up (`float += double`), so the value a word is registered with
 *                                              depends on everything explored before it in the same (node, pos) explorer
 *   src/core/hash/CustomHash_v4_FastUtil81.java:73-89  addTuple: per (k-mer, original branch id) keep the largest PP*
 */
#ifndef RAPPAS_BUILD_ORACLE_H
#define RAPPAS_BUILD_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    int alphabet;   /* 4 or 20: word -> integer code as in ro_kmer_code */
    int k;
    int n_nodes;    /* tested nodes (rows of the posterior table) */
    int n_sites;    /* alignment length L == parsedProbas.getSiteCount() */
    int n_states;   /* parsedProbas.getStateCount() */
    const uint8_t *states;       /* [n_nodes][n_sites][n_states] state at rank j (PProbasSorted.states) */
    const float *pp;             /* [n_nodes][n_sites][n_states] log10 posterior at rank j, descending (PProbasSorted.pp) */
    const uint16_t *node_branch; /* [n_nodes] original branch id the node's k-mers are registered under (WordExplorer_v3.java:93-94) */
    float thr_log10;             /* session.PPStarThresholdAsLog10 */
    int do_gap_jumps;            /* gapJumpsActivated (Main_DBBUILD_3.java:239-258) */
    int limit_to_1_jump;         /* ArgumentsParser_v2.java:76 */
    const uint32_t *gap_off;     /* [n_sites + 1] CSR over sites: Alignment.getGapIntervals()[i] = gap_len[gap_off[i] .. gap_off[i+1]) */
    const int32_t *gap_len;
} ro_build_desc;

typedef struct {
    uint64_t n_keys;
    uint64_t *key_codes;   /* ascending */
    uint64_t *row_offsets; /* [n_keys + 1] */
    uint16_t *branch_ids;  /* ascending inside a row */
    float *scores;
    uint64_t tuples;        /* addTuple calls ("Tuples explored", Main_DBBUILD_3.java:760) */
    uint64_t visits;        /* exploreWords calls that passed the alignment-limit test */
} ro_built;

int ro_build_db(const ro_build_desc *d, ro_built *out);
void ro_built_free(ro_built *b);

#ifdef __cplusplus
}
#endif
#endif
