/*
 * rappas_oracle.c -- CPU restatement of RAPPAS's query-placement hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see rappas_oracle.h).  PARITY UNPINNED: no reference tests / golden
 * vectors exist and the Java reference cannot run here; pinned by hand-derived KATs only.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -fPIC -shared (see oracle/Makefile).
 * All float32 arithmetic below is written one IEEE operation per statement so that the
 * rounding sequence is exactly the Java one (Java float ops round to binary32 after every op).
 */
#include "rappas_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------
 * Alphabets
 * ---------------------------------------------------------------------------------------- */

/* DNA ambiguity classes, alternatives in the order the reference fills them
 * (core/DNAStatesShifted.java:62-96; states A=0 T=1 C=2 G=3, :182-209).
 * '.' and '-' are allocated as byte[4] and never filled => {0,0,0,0} (:57-58). */
enum { DNA_R, DNA_Y, DNA_S, DNA_W, DNA_K, DNA_M, DNA_B, DNA_D, DNA_H, DNA_V, DNA_N, DNA_GAP, DNA_NCLS };
static const uint8_t dna_alt_n[DNA_NCLS] = {2, 2, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4};
static const uint8_t dna_alt[DNA_NCLS][4] = {
    {0, 3},       /* R: A,G */
    {2, 1},       /* Y: C,T */
    {2, 3},       /* S: C,G */
    {0, 1},       /* W: A,T */
    {3, 1},       /* K: G,T */
    {0, 2},       /* M: A,C */
    {2, 3, 1},    /* B: C,G,T */
    {0, 3, 1},    /* D: A,G,T */
    {0, 2, 1},    /* H: A,C,T */
    {0, 2, 3},    /* V: A,C,G */
    {0, 2, 3, 1}, /* N: A,C,G,T */
    {0, 0, 0, 0}, /* '.', '-' : never filled in the reference */
};

/* AA ambiguity classes (core/AAStates.java:97-112): '-','*','!','X','x' -> all 20 in index order;
 * B -> {D,N}={3,7}; Z -> {E,Q}={4,8}; J -> {I,L}={13,14}. */
enum { AA_ANY, AA_B, AA_Z, AA_J, AA_NCLS };

static uint8_t dna_code(uint8_t c) {
    switch (c) { /* core/DNAStatesShifted.java:182-209 (charToByte) and :45-58 (ambiguity keys) */
    case 'A': case 'a': return 0;
    case 'T': case 't': case 'U': case 'u': return 1;
    case 'C': case 'c': return 2;
    case 'G': case 'g': return 3;
    case 'R': case 'r': return 0x80 | DNA_R;
    case 'Y': case 'y': return 0x80 | DNA_Y;
    case 'S': case 's': return 0x80 | DNA_S;
    case 'W': case 'w': return 0x80 | DNA_W;
    case 'K': case 'k': return 0x80 | DNA_K;
    case 'M': case 'm': return 0x80 | DNA_M;
    case 'B': case 'b': return 0x80 | DNA_B;
    case 'D': case 'd': return 0x80 | DNA_D;
    case 'H': case 'h': return 0x80 | DNA_H;
    case 'V': case 'v': return 0x80 | DNA_V;
    case 'N': case 'n': return 0x80 | DNA_N;
    case '.': case '-': return 0x80 | DNA_GAP;
    default: return 0xFF;
    }
}

static uint8_t aa_code(uint8_t c, int convert_uo) {
    /* core/AAStates.java:23-28 (state order R H K D E S T N Q C G P A I L M F W Y V), :68-87 (both cases) */
    static const char order[] = "RHKDESTNQCGPAILMFWYV";
    uint8_t up = (c >= 'a' && c <= 'z') ? (uint8_t)(c - 32) : c;
    /* ambiguity keys first: isAmbiguous() is tested before stateToByte (AmbigSequenceKnife.java:106) */
    switch (c) {
    case '-': case '*': case '!': case 'X': case 'x': return 0x80 | AA_ANY;
    case 'B': case 'b': return 0x80 | AA_B;
    case 'Z': case 'z': return 0x80 | AA_Z;
    case 'J': case 'j': return 0x80 | AA_J;
    default: break;
    }
    for (int i = 0; i < 20; i++)
        if (up == (uint8_t)order[i]) return (uint8_t)i;
    if (convert_uo) { /* core/AAStates.java:118-123 */
        if (up == 'U') return 9;
        if (up == 'O') return 14;
    }
    return 0xFF;
}

uint8_t ro_char_code(int alphabet, int convert_uo, uint8_t c) {
    return alphabet == RO_ALPHABET_DNA ? dna_code(c) : aa_code(c, convert_uo);
}

int ro_amb_alternatives(int alphabet, uint8_t cls, uint8_t *out) {
    if (alphabet == RO_ALPHABET_DNA) {
        if (cls >= DNA_NCLS) return 0;
        memcpy(out, dna_alt[cls], dna_alt_n[cls]);
        return dna_alt_n[cls];
    }
    switch (cls) {
    case AA_ANY: for (int i = 0; i < 20; i++) out[i] = (uint8_t)i; return 20;
    case AA_B: out[0] = 3; out[1] = 7; return 2;
    case AA_Z: out[0] = 4; out[1] = 8; return 2;
    case AA_J: out[0] = 13; out[1] = 14; return 2;
    default: return 0;
    }
}

/* main_v2/Main_DBBUILD_3.java:165-166
 *   float PPStarThreshold=(float)Math.pow((0.0+omega/s.getNonAmbiguousStatesCount()),k);
 *   float PPStarThresholdAsLog=(float)Math.log10(PPStarThreshold);
 * omega is a float (ArgumentsParser_v2.java:52), so omega/nStates is a float division. */
void ro_thresholds(float omega, int n_states, int k, float *thr, float *thr_log10) {
    float ratio = omega / (float)n_states;
    float p = (float)pow(0.0 + (double)ratio, (double)k);
    *thr = p;
    *thr_log10 = (float)log10((double)p);
}

/* core/algos/AmbigSequenceKnife.java:95 */
int ro_max_ambig_per_mer(int k, int n_states) {
    return (int)floor(pow((double)k, 1.0 / (double)n_states));
}

/* core/DNAStatesShifted.java:115-143, byte-wise: base i goes to bits 2*(i%4) of byte i/4 */
int ro_compress_mer_dna(const uint8_t *states, int k, uint8_t *out) {
    int byte_count = (int)ceil((0.0 + k) / 4);
    uint8_t four = 0;
    for (int i = 0; i < k; i++) {
        if ((i > 0) & (i % 4 == 0)) {
            out[(i / 4) - 1] = four;
            four = 0;
        }
        four = (uint8_t)(four | (states[i] << (2 * (i % 4))));
    }
    out[byte_count - 1] = four;
    return byte_count;
}

uint64_t ro_kmer_code(int alphabet, const uint8_t *states, int k) {
    uint64_t code = 0;
    if (alphabet == RO_ALPHABET_DNA) {
        uint8_t bytes[16];
        int n = ro_compress_mer_dna(states, k, bytes);
        for (int i = 0; i < n; i++) code |= (uint64_t)bytes[i] << (8 * i);
    } else {
        for (int i = 0; i < k; i++) code |= (uint64_t)states[i] << (5 * i);
    }
    return code;
}

/* ------------------------------------------------------------------------------------------
 * DB: code -> CSR row (membership + row iteration order are all the reference's map contributes:
 * core/hash/CustomHash_v4_FastUtil81.java:146-153, HashStrategy.java:22-29)
 * ---------------------------------------------------------------------------------------- */
struct ro_db {
    int alphabet, convert_uo, k, n_branches;
    float T, P; /* PPStarThresholdAsLog10, PPStarThreshold (main_v2/SessionNext_v2.java:43-66) */
    uint64_t n_keys, n_entries;
    uint64_t *row_off;
    uint16_t *branch;
    float *score;
    uint64_t cap;     /* power of two */
    uint64_t *slot_key; /* code+1, 0 = empty */
    uint64_t *slot_row;
};

static uint64_t mix64(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return x;
}

ro_db *ro_db_create(int alphabet, int convert_uo, int k, int n_branches, float thr_log10, float thr,
                    uint64_t n_keys, const uint64_t *key_codes, const uint64_t *row_offsets,
                    const uint16_t *branch_ids, const float *scores) {
    if (alphabet != RO_ALPHABET_DNA && alphabet != RO_ALPHABET_AA) return NULL;
    if (k < 1 || (alphabet == RO_ALPHABET_DNA && k > 31) || (alphabet == RO_ALPHABET_AA && k > 12)) return NULL;
    ro_db *db = (ro_db *)calloc(1, sizeof(ro_db));
    if (!db) return NULL;
    db->alphabet = alphabet; db->convert_uo = convert_uo; db->k = k; db->n_branches = n_branches;
    db->T = thr_log10; db->P = thr; db->n_keys = n_keys;
    db->n_entries = n_keys ? row_offsets[n_keys] : 0;
    db->row_off = (uint64_t *)malloc((n_keys + 1) * sizeof(uint64_t));
    db->branch = (uint16_t *)malloc((db->n_entries + 1) * sizeof(uint16_t));
    db->score = (float *)malloc((db->n_entries + 1) * sizeof(float));
    uint64_t cap = 16;
    while (cap < 2 * n_keys) cap <<= 1;
    db->cap = cap;
    db->slot_key = (uint64_t *)calloc(cap, sizeof(uint64_t));
    db->slot_row = (uint64_t *)calloc(cap, sizeof(uint64_t));
    if (!db->row_off || !db->branch || !db->score || !db->slot_key || !db->slot_row) { ro_db_destroy(db); return NULL; }
    if (n_keys) memcpy(db->row_off, row_offsets, (n_keys + 1) * sizeof(uint64_t)); else db->row_off[0] = 0;
    memcpy(db->branch, branch_ids, db->n_entries * sizeof(uint16_t));
    memcpy(db->score, scores, db->n_entries * sizeof(float));
    for (uint64_t r = 0; r < n_keys; r++) {
        uint64_t h = mix64(key_codes[r]) & (cap - 1);
        while (db->slot_key[h]) {
            if (db->slot_key[h] == key_codes[r] + 1) { ro_db_destroy(db); return NULL; } /* duplicate key */
            h = (h + 1) & (cap - 1);
        }
        db->slot_key[h] = key_codes[r] + 1;
        db->slot_row[h] = r;
    }
    return db;
}

void ro_db_destroy(ro_db *db) {
    if (!db) return;
    free(db->row_off); free(db->branch); free(db->score); free(db->slot_key); free(db->slot_row);
    free(db);
}

/* hash.get(key): returns row index or -1 (CustomHash_v4_FastUtil81.java:146-153) */
static int64_t db_get(const ro_db *db, uint64_t code) {
    uint64_t h = mix64(code) & (db->cap - 1);
    while (db->slot_key[h]) {
        if (db->slot_key[h] == code + 1) return (int64_t)db->slot_row[h];
        h = (h + 1) & (db->cap - 1);
    }
    return -1;
}

/* ------------------------------------------------------------------------------------------
 * java.util.PriorityQueue<Score> restated (Comparable path), Score.compareTo = Float.compare
 * (PlacementProcess.java:1247-1262)
 * ---------------------------------------------------------------------------------------- */
typedef struct { int32_t node; float score; } score_t;

static int float_compare(float a, float b) { /* java.lang.Float.compare */
    if (a < b) return -1;
    if (a > b) return 1;
    int32_t ia, ib;
    memcpy(&ia, &a, 4); memcpy(&ib, &b, 4);
    if (a != a) ia = 0x7fc00000; /* floatToIntBits canonical NaN */
    if (b != b) ib = 0x7fc00000;
    return ia == ib ? 0 : (ia < ib ? -1 : 1);
}

static void pq_sift_up(score_t *q, int k, score_t x) {
    while (k > 0) {
        int parent = (k - 1) >> 1;
        score_t e = q[parent];
        if (float_compare(x.score, e.score) >= 0) break;
        q[k] = e;
        k = parent;
    }
    q[k] = x;
}

static void pq_sift_down(score_t *q, int n, int k, score_t x) {
    int half = n >> 1;
    while (k < half) {
        int child = (k << 1) + 1;
        score_t c = q[child];
        int right = child + 1;
        if (right < n && float_compare(c.score, q[right].score) > 0) c = q[child = right];
        if (float_compare(x.score, c.score) <= 0) break;
        q[k] = c;
        k = child;
    }
    q[k] = x;
}

static void pq_add(score_t *q, int *size, score_t e) {
    int i = (*size)++;
    if (i == 0) q[0] = e; else pq_sift_up(q, i, e);
}

static void pq_poll(score_t *q, int *size) {
    int n = --(*size);
    score_t x = q[n];
    if (n > 0) pq_sift_down(q, n, 0, x);
}

/* ------------------------------------------------------------------------------------------
 * Per-read work area (allocated once per batch, like PlacementProcess.java:493-501)
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    int32_t *L; int32_t L_size;
    int32_t *C; float *S;
    int32_t *L_amb; int32_t *C_amb; float *S_amb;
    int8_t *seq; uint8_t *cls; int32_t *amb_count; uint64_t seq_cap;
    score_t *heap; score_t *best; int keep_cap;
    float *tie_tmp;
} work_t;

static int work_init(work_t *w, int n_branches, int keep_at_most) {
    memset(w, 0, sizeof(*w));
    w->L = (int32_t *)malloc(sizeof(int32_t) * (size_t)n_branches);
    w->C = (int32_t *)calloc((size_t)n_branches, sizeof(int32_t));
    w->S = (float *)calloc((size_t)n_branches, sizeof(float));
    w->L_amb = (int32_t *)malloc(sizeof(int32_t) * (size_t)n_branches);
    w->C_amb = (int32_t *)calloc((size_t)n_branches, sizeof(int32_t));
    w->S_amb = (float *)calloc((size_t)n_branches, sizeof(float));
    w->heap = (score_t *)malloc(sizeof(score_t) * (size_t)(keep_at_most + 2));
    w->best = (score_t *)malloc(sizeof(score_t) * (size_t)(keep_at_most + 2));
    w->tie_tmp = (float *)malloc(sizeof(float) * (size_t)n_branches);
    w->keep_cap = keep_at_most;
    return (w->L && w->C && w->S && w->L_amb && w->C_amb && w->S_amb && w->heap && w->best && w->tie_tmp) ? 0 : -1;
}

static void work_free(work_t *w) {
    free(w->L); free(w->C); free(w->S); free(w->L_amb); free(w->C_amb); free(w->S_amb);
    free(w->seq); free(w->cls); free(w->amb_count); free(w->heap); free(w->best); free(w->tie_tmp);
}

static int work_reserve_seq(work_t *w, uint64_t R) {
    if (R <= w->seq_cap) return 0;
    free(w->seq); free(w->cls); free(w->amb_count);
    w->seq_cap = R + 64;
    w->seq = (int8_t *)malloc(w->seq_cap);
    w->cls = (uint8_t *)malloc(w->seq_cap);
    w->amb_count = (int32_t *)malloc(w->seq_cap * sizeof(int32_t));
    return (w->seq && w->cls && w->amb_count) ? 0 : -1;
}

/* A6: accumulate one row.  PlacementProcess.java:719-735 */
static void accumulate_row(const ro_db *db, work_t *w, int64_t row, int Q, uint64_t *entries) {
    const float T = db->T;
    for (uint64_t e = db->row_off[row]; e < db->row_off[row + 1]; e++) {
        int x = (int)db->branch[e];
        float v = db->score[e];
        if (w->C[x] == 0) {
            w->L[w->L_size++] = x;
            float qt = (float)Q * T;  /* int*float -> float multiply (:728) */
            w->S[x] = w->S[x] + qt;   /* S[x] += ... (:728) */
        }
        w->C[x] += 1;
        float d = v - T;              /* float subtract (:733) */
        w->S[x] = w->S[x] + d;        /* float add (:733) */
        (*entries)++;
    }
}

/* A7 mean: PlacementProcess.java:1129-1174 */
static void ambiguous_mean(const ro_db *db, work_t *w, const int64_t *rows, int W, int Q, uint64_t *entries) {
    const float T = db->T, P = db->P;
    int32_t la = 0;
    for (int i = 0; i < W; i++) {
        if (rows[i] < 0) continue;
        for (uint64_t e = db->row_off[rows[i]]; e < db->row_off[rows[i] + 1]; e++) {
            int x = (int)db->branch[e];
            float v = db->score[e];
            if (w->C_amb[x] == 0) w->L_amb[la++] = x;
            w->C_amb[x] += 1;
            /* S_amb[x]+=Math.pow(10,entry.getFloatValue());  float += double  (:1155) */
            w->S_amb[x] = (float)((double)w->S_amb[x] + pow(10.0, (double)v));
            (*entries)++;
        }
    }
    for (int32_t i = 0; i < la; i++) {
        int x = w->L_amb[i];
        if (w->C[x] == 0) {
            w->L[w->L_size++] = x;
            w->S[x] = (float)Q * T; /* :1165 */
        }
        w->C[x] += 1;
        /* float avgProba=(S_amb[x] + (W_size-C_amb[x])*PPStarThreshold) / W_size;  all float (:1168) */
        float missing = (float)(W - w->C_amb[x]);
        float pad = missing * P;
        float tot = w->S_amb[x] + pad;
        float avg = tot / (float)W;
        /* S[x]+=Math.log10(avgProba)-PPStarThresholdAsLog10;  float += double (:1169) */
        w->S[x] = (float)((double)w->S[x] + (log10((double)avg) - (double)T));
        w->C_amb[x] = 0;
        w->S_amb[x] = 0.0f; /* the reference allocates fresh zeroed arrays per ambiguous k-mer (:1130-1131) */
    }
}

/* A7 max: PlacementProcess.java:1185-1236 */
static void ambiguous_max(const ro_db *db, work_t *w, const int64_t *rows, int W, int Q, uint64_t *entries) {
    const float T = db->T;
    int32_t la = 0;
    for (int i = 0; i < W; i++) {
        if (rows[i] < 0) continue;
        for (uint64_t e = db->row_off[rows[i]]; e < db->row_off[rows[i] + 1]; e++) {
            int x = (int)db->branch[e];
            float v = db->score[e];
            if (w->C_amb[x] == 0) { w->L_amb[la++] = x; w->S_amb[x] = v; }
            w->C_amb[x] += 1;
            if (v > w->S_amb[x]) w->S_amb[x] = v;
            (*entries)++;
        }
    }
    for (int32_t i = 0; i < la; i++) {
        int x = w->L_amb[i];
        if (w->C[x] == 0) {
            w->L[w->L_size++] = x;
            w->S[x] = (float)Q * T; /* :1226 */
        }
        w->C[x] += 1;
        float d = w->S_amb[x] - T;
        w->S[x] = w->S[x] + d;     /* :1230 */
        w->C_amb[x] = 0;
        w->S_amb[x] = 0.0f;
    }
}

/* A1-A7: knife + lookup + accumulate for one read. Returns flags (without PLACED). */
static uint32_t score_read(const ro_db *db, work_t *w, int amb_mode, const uint8_t *s, uint64_t R,
                           uint64_t *entries, ro_counters *ct) {
    const int k = db->k;
    uint32_t flags = 0;
    w->L_size = 0;
    if (work_reserve_seq(w, R)) return RO_FLAG_BAD_CHAR;
    /* AmbigSequenceKnife.initTables :98-130 */
    for (uint64_t i = 0; i < R; i++) w->amb_count[i] = 0;
    for (uint64_t i = 0; i < R; i++) {
        uint8_t c = ro_char_code(db->alphabet, db->convert_uo, s[i]);
        if (c == 0xFF) { flags |= RO_FLAG_BAD_CHAR; w->seq[i] = 0; w->cls[i] = 0; continue; }
        if (c & 0x80) {
            flags |= RO_FLAG_AMBIGUOUS;
            int64_t j0 = (int64_t)i - k + 1;
            for (int64_t j = j0; j < (int64_t)i + 1; j++)
                if (j > -1 && j < (int64_t)R) w->amb_count[j]++;
            w->seq[i] = -1;
            w->cls[i] = (uint8_t)(c & 0x7F);
        } else {
            w->seq[i] = (int8_t)c;
        }
    }
    if (flags & RO_FLAG_BAD_CHAR) /* reference: System.exit(1) (:124-128); both conditions are reported */
        return flags | (R < (uint64_t)k ? RO_FLAG_TOO_SHORT : 0u);
    if (R < (uint64_t)k) return flags | RO_FLAG_TOO_SHORT; /* merOrder length R-k+1 <= 0 (:145) */
    const int Q = (int)(R - (uint64_t)k + 1);   /* sk.getMerCount() (:191) */
    const int max_amb = ro_max_ambig_per_mer(k, db->alphabet);
    uint8_t window[32];
    for (int j = 0; j < Q; j++) { /* SAMPLING_LINEAR (:144-150); getNextByteWord (:210-272) */
        ct->kmers++;
        if (w->amb_count[j] < 1) {
            for (int i = 0; i < k; i++) window[i] = (uint8_t)w->seq[j + i];
            int64_t row = db_get(db, ro_kmer_code(db->alphabet, window, k)); /* PlacementProcess.java:704-708 */
            if (row < 0) continue;                                            /* :713-716 */
            ct->kmers_hit++;
            accumulate_row(db, w, row, Q, entries);
        } else if (w->amb_count[j] > max_amb) {
            ct->skipped_kmers++; /* byte[1] sentinel (:229-232) -> PlacementProcess.java:691-696 */
        } else if (amb_mode == RO_AMB_SKIP) {
            ct->skipped_kmers++; /* --noamb: PlacementProcess.java:745-749 */
        } else {
            /* 1 .. max_amb ambiguous positions (max_amb = 1 for DNA k < 16 and every protein k, 2 for DNA k = 16 .. 80).
             * AmbigSequenceKnife.java:235-260: altProduct = product of the alternative counts; word j takes, at EVERY ambiguous
             * position i, alt_i[j mod len_i] (the `jump` counter runs 0 .. altProduct-1 for each position on its own) -- the
             * cartesian product only when the counts are coprime; with two positions of equal count the same words come
             * up several times, and PlacementProcess counts each occurrence (W_size = words.length / k, :1133). */
            ct->amb_kmers++;
            int pos[4], np = 0;
            for (int i = 0; i < k; i++) { window[i] = (uint8_t)w->seq[j + i]; if (w->seq[j + i] == -1 && np < 4) pos[np++] = i; }
            uint8_t alts[4][20];
            int nalt[4], W = 1;
            for (int q = 0; q < np; q++) { nalt[q] = ro_amb_alternatives(db->alphabet, w->cls[j + pos[q]], alts[q]); W *= nalt[q]; }
            if (W > 400) { ct->skipped_kmers++; continue; } /* (unreachable: <= 2 positions for every k this oracle accepts) */
            static __thread int64_t rows[400];
            int any = 0;
            for (int a = 0; a < W; a++) { /* words[i+jump*k]=alt_i[jump % len_i] (:249-256) */
                for (int q = 0; q < np; q++) window[pos[q]] = alts[q][a % nalt[q]];
                rows[a] = db_get(db, ro_kmer_code(db->alphabet, window, k));
                if (rows[a] >= 0) any = 1;
            }
            if (any) ct->kmers_hit++;
            if (amb_mode == RO_AMB_MAX) ambiguous_max(db, w, rows, W, Q, entries);
            else ambiguous_mean(db, w, rows, W, Q, entries);
        }
    }
    return flags;
}

static void reset_scores(work_t *w) { /* PlacementProcess.java:1067-1075 */
    for (int32_t i = 0; i < w->L_size; i++) { w->S[w->L[i]] = 0.0f; w->C[w->L[i]] = 0; }
    w->L_size = 0;
}

static int cmp_float_desc(const void *a, const void *b) {
    float x = *(const float *)a, y = *(const float *)b;
    return (x < y) - (x > y);
}

/* A8 + A9 for one read whose S/L are filled. Writes rows best->worse. Returns n_rows. */
static int select_and_weigh(work_t *w, int K, float keep_factor, float ns_bound, uint16_t *o_branch,
                            float *o_score, double *o_lwr, uint32_t *flags) {
    const int nL = w->L_size;
    int numBest = K;
    if (nL < K) numBest = nL; /* PlacementProcess.java:828-832 */

    /* fillBestScoreList :396-451 */
    int hs = 0;
    for (int i = 0; i < nL; i++) {
        score_t e = {w->L[i], w->S[w->L[i]]};
        pq_add(w->heap, &hs, e);
        if (hs > numBest) pq_poll(w->heap, &hs);
    }
    double sum = 0.0;
    float lowest = 0.0f, best = -3.4028234663852886e38f; /* -Float.MAX_VALUE */
    for (int i = 0; i < hs; i++) {
        sum += pow(10.0, (double)w->heap[i].score); /* :418 */
        if (w->heap[i].score < lowest) lowest = w->heap[i].score;
        if (w->heap[i].score > best) best = w->heap[i].score;
    }
    /* bestScoreList: first hs slots overwritten in heap-array order, rest stay (-1,-inf) placeholders */
    for (int i = 0; i < K; i++) { w->best[i].node = -1; w->best[i].score = -INFINITY; }
    for (int i = 0; i < hs; i++) w->best[i] = w->heap[i];
    /* Arrays.sort(Object[]): stable ascending (:436) -> stable insertion sort */
    for (int i = 1; i < K; i++) {
        score_t x = w->best[i];
        int j = i - 1;
        while (j >= 0 && float_compare(w->best[j].score, x.score) > 0) { w->best[j + 1] = w->best[j]; j--; }
        w->best[j + 1] = x;
    }
    float shift = (-308.0f >= lowest) ? best : 0.0f; /* computeWeightRatioShift :384-390 */
    if (shift != 0.0f) {
        sum = 0.0;
        for (int ii = K - numBest; ii < K; ii++) {
            float d = w->best[ii].score - shift; /* float subtract, then widen (:445-447) */
            sum += pow(10.0, (double)d);
        }
    }

    /* tie diagnostic (oracle only): exact equality among the top-(K+1) touched scores */
    for (int i = 0; i < nL; i++) w->tie_tmp[i] = w->S[w->L[i]];
    qsort(w->tie_tmp, (size_t)nL, sizeof(float), cmp_float_desc);
    int lim = nL < K + 1 ? nL : K + 1;
    for (int i = 1; i < lim; i++)
        if (w->tie_tmp[i] == w->tie_tmp[i - 1]) *flags |= RO_FLAG_TIE;

    /* :974 gate */
    if (!(w->best[K - 1].score >= ns_bound)) { *flags |= RO_FLAG_BELOW_NSBOUND; return 0; }
    float best2 = w->best[K - 1].score;
    float lowest2 = w->best[K - numBest].score;
    float shift2 = (-308.0f >= lowest2) ? best2 : 0.0f; /* :978-980 */
    double bestRatio = -1;
    int n = 0;
    for (int i = K - 1; i > K - numBest - 1; i--) { /* :984 */
        /* computeWeightRatio :392-394 : float - double => double subtraction */
        double ratio = pow(10.0, (double)w->best[i].score - (double)shift2) / sum;
        if (i == K - 1) bestRatio = ratio;
        if (i < K - 1 && ratio < (bestRatio * (double)keep_factor)) break; /* :998-1000 */
        o_branch[n] = (uint16_t)w->best[i].node;
        o_score[n] = w->best[i].score;
        o_lwr[n] = ratio;
        n++;
    }
    return n;
}

int ro_place_batch(const ro_db *db, int keep_at_most, float keep_factor, int amb_mode, float ns_bound,
                   uint64_t n_reads, const uint8_t *seq, const uint64_t *seq_off, uint8_t *n_rows,
                   uint16_t *branch, float *score, double *lwr, uint32_t *flags,
                   uint32_t *entries_per_read, ro_counters *counters) {
    if (!db || keep_at_most < 1 || keep_at_most > 255) return -1;
    work_t w;
    if (work_init(&w, db->n_branches, keep_at_most)) { work_free(&w); return -2; }
    ro_counters ct;
    memset(&ct, 0, sizeof(ct));
    const int K = keep_at_most;
    for (uint64_t r = 0; r < n_reads; r++) {
        const uint8_t *s = seq + seq_off[r];
        uint64_t R = seq_off[r + 1] - seq_off[r];
        uint64_t H = 0;
        for (int i = 0; i < K; i++) { branch[r * K + i] = 0xFFFF; score[r * K + i] = -INFINITY; lwr[r * K + i] = 0.0; }
        uint32_t f = score_read(db, &w, amb_mode, s, R, &H, &ct);
        int n = 0;
        if (w.L_size > 0) { /* PlacementProcess.java:797 */
            f |= RO_FLAG_PLACED;
            n = select_and_weigh(&w, K, keep_factor, ns_bound, branch + r * K, score + r * K, lwr + r * K, &f);
            ct.placed++;
        } else {
            ct.unplaced++;
        }
        n_rows[r] = (uint8_t)n;
        flags[r] = f;
        if (entries_per_read) entries_per_read[r] = (uint32_t)H;
        ct.entries += H;
        ct.reads++;
        reset_scores(&w);
    }
    if (counters) *counters = ct;
    work_free(&w);
    return 0;
}

int ro_score_vector(const ro_db *db, int amb_mode, const uint8_t *seq, uint64_t len, float *S_out,
                    int32_t *L_out, int32_t *L_size) {
    work_t w;
    if (work_init(&w, db->n_branches, 1)) { work_free(&w); return -2; }
    ro_counters ct;
    memset(&ct, 0, sizeof(ct));
    uint64_t H = 0;
    uint32_t f = score_read(db, &w, amb_mode, seq, len, &H, &ct);
    for (int i = 0; i < db->n_branches; i++) S_out[i] = NAN;
    for (int32_t i = 0; i < w.L_size; i++) { S_out[w.L[i]] = w.S[w.L[i]]; L_out[i] = w.L[i]; }
    *L_size = w.L_size;
    work_free(&w);
    return (int)f;
}
