/* rappas_build_oracle.c -- see rappas_build_oracle.h.  TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED. */
#include "rappas_build_oracle.h"

#include <stdlib.h>
#include <string.h>

typedef struct {
    uint64_t key; /* code << 16 | branch */
    float score;
} tuple_t;

typedef struct {
    const ro_build_desc *d;
    int node;
    uint16_t branch;
    /* WordExplorer_v3 fields (:37-55) */
    float currentLogSum;
    uint8_t word[32];
    int boundReached, boundReachingK, current_k, idxOfFirstJump;
    /* output */
    tuple_t *t;
    uint64_t n, cap, visits;
    int oom;
} explorer_t;

static void add_tuple(explorer_t *e) {
    const ro_build_desc *d = e->d;
    uint64_t code = 0;
    const int bits = d->alphabet == 4 ? 2 : 5; /* compressMer: DNAStatesShifted.java:115-143 / AAStates.java:195-197 */
    for (int i = 0; i < d->k; i++) code |= (uint64_t)e->word[i] << (bits * i);
    if (e->n == e->cap) {
        uint64_t nc = e->cap ? e->cap * 2 : 1 << 16;
        tuple_t *nt = (tuple_t *)realloc(e->t, nc * sizeof(tuple_t));
        if (!nt) { e->oom = 1; return; }
        e->t = nt; e->cap = nc;
    }
    e->t[e->n].key = (code << 16) | e->branch;
    e->t[e->n].score = e->currentLogSum;
    e->n++;
}

/* WordExplorer_v3.java:98-199, statement for statement */
static void explore_words(explorer_t *e, int i, int j) {
    const ro_build_desc *d = e->d;
    if (i > d->n_sites - 1) return;                                            /* :109-111 */
    if (e->current_k == 0) e->idxOfFirstJump = -1;                             /* :113-115 */
    const size_t at = ((size_t)e->node * d->n_sites + (size_t)i) * d->n_states + (size_t)j;
    e->visits++;
    e->word[e->current_k] = d->states[at];                                     /* :117 */
    const double pp = (double)d->pp[at];                                       /* getPP returns double (PProbasSorted.java:45) */
    e->currentLogSum = (float)((double)e->currentLogSum + pp);                 /* :119  float += double */
    e->boundReached = e->currentLogSum < d->thr_log10;                         /* :120 */
    if (e->boundReached) e->boundReachingK = e->current_k;                     /* :121-123 */
    if (e->current_k == d->k - 1) {                                            /* :126 */
        if (!e->boundReached) add_tuple(e);                                    /* :128-138 */
        e->currentLogSum = (float)((double)e->currentLogSum - pp);             /* :141 */
        return;
    }
    for (int j2 = 0; j2 < d->n_states; j2++) {                                 /* :147 */
        if (e->boundReached && e->boundReachingK == e->current_k + 1) break;   /* :148-150 */
        e->current_k++;
        explore_words(e, i + 1, j2);                                           /* :155-157 */
        e->current_k--;
        if (d->do_gap_jumps && i < d->n_sites - 1) {                           /* :161 */
            const uint32_t g0 = d->gap_off[i + 1], g1 = d->gap_off[i + 2];
            if (g1 > g0) {                                                     /* :163 gapIntervals[i+1] != null */
                if (!d->limit_to_1_jump) {                                     /* :165-171 */
                    for (uint32_t g = g0; g < g1; g++) {
                        e->current_k++;
                        explore_words(e, (i + 1) + d->gap_len[g], j2);
                        e->current_k--;
                    }
                } else if (e->idxOfFirstJump == -1) {                          /* :174-184 */
                    e->idxOfFirstJump = i;
                    for (uint32_t g = g0; g < g1; g++) {
                        e->current_k++;
                        explore_words(e, (i + 1) + d->gap_len[g], j2);
                        e->current_k--;
                    }
                }
            }
        }
    }
    e->currentLogSum = (float)((double)e->currentLogSum - pp);                 /* :198 */
}

static int cmp_tuple(const void *a, const void *b) {
    const tuple_t *x = (const tuple_t *)a, *y = (const tuple_t *)b;
    if (x->key != y->key) return x->key < y->key ? -1 : 1;
    return (x->score > y->score) ? -1 : (x->score < y->score ? 1 : 0); /* best first */
}

int ro_build_db(const ro_build_desc *d, ro_built *out) {
    memset(out, 0, sizeof(*out));
    if (!d || d->k < 1 || d->k > 31 || d->n_states < 1 || d->n_sites < 1 || d->n_nodes < 0) return -1;
    explorer_t e;
    memset(&e, 0, sizeof(e));
    e.d = d;
    for (int node = 0; node < d->n_nodes; node++) {                            /* Main_DBBUILD_3.java:648 */
        for (int pos = 0; pos < d->n_sites - d->k + 2; pos++) {                /* :693 */
            /* new WordExplorer_v3(...) :697-704 */
            e.node = node;
            e.branch = d->node_branch[node];
            e.currentLogSum = 0.0f;
            e.boundReached = 0;
            e.boundReachingK = -1;
            e.current_k = 0;
            e.idxOfFirstJump = -1;
            for (int j = 0; j < d->n_states; j++) explore_words(&e, pos, j);   /* :710-712 */
            if (e.oom) { free(e.t); return -2; }
        }
    }
    out->tuples = e.n;
    out->visits = e.visits;
    /* CustomHash_v4_FastUtil81.addTuple (:73-89): the largest PP* per (word, nodeId) */
    if (e.n) qsort(e.t, e.n, sizeof(tuple_t), cmp_tuple);
    uint64_t n_ent = 0, n_keys = 0;
    for (uint64_t i = 0; i < e.n; i++) {
        if (i == 0 || e.t[i].key != e.t[i - 1].key) {
            n_ent++;
            if (i == 0 || (e.t[i].key >> 16) != (e.t[i - 1].key >> 16)) n_keys++;
        }
    }
    out->n_keys = n_keys;
    out->key_codes = (uint64_t *)malloc((n_keys + 1) * sizeof(uint64_t));
    out->row_offsets = (uint64_t *)malloc((n_keys + 1) * sizeof(uint64_t));
    out->branch_ids = (uint16_t *)malloc((n_ent + 1) * sizeof(uint16_t));
    out->scores = (float *)malloc((n_ent + 1) * sizeof(float));
    if (!out->key_codes || !out->row_offsets || !out->branch_ids || !out->scores) { free(e.t); ro_built_free(out); return -2; }
    uint64_t ki = 0, ei = 0;
    for (uint64_t i = 0; i < e.n; i++) {
        if (i && e.t[i].key == e.t[i - 1].key) continue;
        if (i == 0 || (e.t[i].key >> 16) != (e.t[i - 1].key >> 16)) {
            out->key_codes[ki] = e.t[i].key >> 16;
            out->row_offsets[ki] = ei;
            ki++;
        }
        out->branch_ids[ei] = (uint16_t)(e.t[i].key & 0xFFFF);
        out->scores[ei] = e.t[i].score;
        ei++;
    }
    out->row_offsets[ki] = ei;
    free(e.t);
    return 0;
}

void ro_built_free(ro_built *b) {
    if (!b) return;
    free(b->key_codes); free(b->row_offsets); free(b->branch_ids); free(b->scores);
    memset(b, 0, sizeof(*b));
}
