/*
 * rappas_jni.c -- thin JNI adapter over the C ABI of include/rappas_place.h.
 *
 * Java side: class core.algos.NativePlacement.  This file is compiled only when a JDK is
 * available (JAVA_HOME set): this image has no JVM and no jni.h, so it is NOT part of build()/tests and has never run
 * (UNTESTED; only syntax-checked with gcc -fsyntax-only against a throwaway declaration of the JNI entry points it uses).
 * Build:
 *   gcc -shared -fPIC -I$JAVA_HOME/include -I$JAVA_HOME/include/linux -Iinclude \
 *       rappas_amd/csrc/jni/rappas_jni.c -Lrappas_amd -lrappas_place -o librappas_jni.so
 */
#include <jni.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "rappas_place.h"

static void throw_new(JNIEnv *env, const char *cls, const char *msg) {
    jclass ex = (*env)->FindClass(env, cls);
    if (ex) (*env)->ThrowNew(env, ex, msg);
}
#define THROW_ARG(msg) throw_new(env, "java/lang/IllegalArgumentException", msg)
#define THROW_OOM(msg) throw_new(env, "java/lang/OutOfMemoryError", msg)

static void throw_rk(JNIEnv *env, const char *where) {
    char msg[640];
    snprintf(msg, sizeof msg, "%s: %s", where, rk_last_error());
    jclass ex = (*env)->FindClass(env, "java/lang/RuntimeException");
    if (ex) (*env)->ThrowNew(env, ex, msg);
}

/* Every array length is checked against the counts the engine will trust BEFORE any native pointer is taken: a wrong-sized
 * Java array must become an IllegalArgumentException, never a native heap overwrite.  Get*ArrayElements results are
 * NULL-checked (the JVM has then already raised OutOfMemoryError). */

/* long dbCreate(int alphabet, boolean convertUO, int k, int nBranches, float thrLog10, float thr,
 *               long[] keyCodes, long[] rowOffsets, char[] branchIds, float[] scores, int device)
 * The Java caller fills the arrays by walking session.hash exactly like SessionNext_v2.saveToJSON
 * (src/main_v2/SessionNext_v2.java:250-261). */
JNIEXPORT jlong JNICALL Java_core_algos_NativePlacement_dbCreate(JNIEnv *env, jclass cls, jint alphabet,
        jboolean convertUO, jint k, jint nBranches, jfloat thrLog10, jfloat thr, jlongArray keyCodes,
        jlongArray rowOffsets, jcharArray branchIds, jfloatArray scores, jint device) {
    (void)cls;
    if (!keyCodes || !rowOffsets || !branchIds || !scores) { THROW_ARG("dbCreate: null array"); return 0; }
    if (k < 0 || nBranches < 0) { THROW_ARG("dbCreate: negative k / nBranches"); return 0; }
    const jsize nKeys = (*env)->GetArrayLength(env, keyCodes);
    const jsize nEnt = (*env)->GetArrayLength(env, branchIds);
    if ((*env)->GetArrayLength(env, rowOffsets) != nKeys + 1) { THROW_ARG("dbCreate: rowOffsets must have keyCodes.length + 1 elements"); return 0; }
    if ((*env)->GetArrayLength(env, scores) != nEnt) { THROW_ARG("dbCreate: branchIds and scores differ in length"); return 0; }
    rk_db_desc d;
    memset(&d, 0, sizeof d);
    d.alphabet = (uint32_t)alphabet; d.convert_uo = convertUO ? 1u : 0u; d.k = (uint32_t)k;
    d.n_branches = (uint32_t)nBranches; d.thr_log10 = thrLog10; d.thr = thr;
    d.n_keys = (uint64_t)nKeys;
    d.device = device; d.table_mode = RK_TABLE_AUTO;
    jlong *kc = (*env)->GetLongArrayElements(env, keyCodes, NULL);
    jlong *ro = kc ? (*env)->GetLongArrayElements(env, rowOffsets, NULL) : NULL;
    jchar *br = ro ? (*env)->GetCharArrayElements(env, branchIds, NULL) : NULL;   /* jchar == uint16: (char)nodeId */
    jfloat *sc = br ? (*env)->GetFloatArrayElements(env, scores, NULL) : NULL;
    rk_db *db = NULL;
    int rc = RK_ERR_NOMEM;
    int bad_offsets = 0;
    if (sc) {
        /* the CSR must end exactly at the entry arrays' length: rk_db_create reads branchIds / scores up to rowOffsets[nKeys] */
        bad_offsets = (ro[0] != 0 || (uint64_t)ro[nKeys] != (uint64_t)nEnt);
        if (!bad_offsets) {
            d.key_codes = (const uint64_t *)kc; d.row_offsets = (const uint64_t *)ro;
            d.branch_ids = (const uint16_t *)br; d.scores = sc;
            rc = rk_db_create(&d, &db);
        }
    }
    if (sc) (*env)->ReleaseFloatArrayElements(env, scores, sc, JNI_ABORT);
    if (br) (*env)->ReleaseCharArrayElements(env, branchIds, br, JNI_ABORT);
    if (ro) (*env)->ReleaseLongArrayElements(env, rowOffsets, ro, JNI_ABORT);
    if (kc) (*env)->ReleaseLongArrayElements(env, keyCodes, kc, JNI_ABORT);
    if (!sc) { if (!(*env)->ExceptionCheck(env)) THROW_OOM("dbCreate: could not pin the arrays"); return 0; }
    if (bad_offsets) { THROW_ARG("dbCreate: rowOffsets must start at 0 and end at branchIds.length"); return 0; }
    if (rc != RK_OK) { throw_rk(env, "rk_db_create"); return 0; }
    return (jlong)(intptr_t)db;
}

/* shared by placeBatch / placeBatchMulti: length checks, pinning, the call, release */
static void place_common(JNIEnv *env, rk_db *const *handles, uint32_t n_handles, jbyteArray seqs, jlongArray offs, jint keepAtMost,
                         jfloat keepFactor, jint ambMode, jfloat nsBound, jbyteArray nRows, jcharArray branch, jfloatArray score,
                         jdoubleArray lwr, jintArray flags) {
    if (!seqs || !offs || !nRows || !branch || !score || !lwr || !flags) { THROW_ARG("placeBatch: null array"); return; }
    if (keepAtMost < 1 || keepAtMost > 16) { THROW_ARG("placeBatch: keepAtMost outside 1..16"); return; }
    const jsize nOff = (*env)->GetArrayLength(env, offs);
    if (nOff < 1) { THROW_ARG("placeBatch: offs needs n + 1 >= 1 elements"); return; }
    const jsize n = nOff - 1;
    const jlong rows = (jlong)n * (jlong)keepAtMost;
    if ((*env)->GetArrayLength(env, nRows) < n || (*env)->GetArrayLength(env, flags) < n ||
        (jlong)(*env)->GetArrayLength(env, branch) < rows || (jlong)(*env)->GetArrayLength(env, score) < rows ||
        (jlong)(*env)->GetArrayLength(env, lwr) < rows) {
        THROW_ARG("placeBatch: result arrays need n (nRows, flags) and n * keepAtMost (branch, score, lwr) elements");
        return;
    }
    const jsize nSeq = (*env)->GetArrayLength(env, seqs);
    rk_params p = {(uint32_t)keepAtMost, keepFactor, (uint32_t)ambMode, nsBound};
    jbyte *s = (*env)->GetByteArrayElements(env, seqs, NULL);
    jlong *o = s ? (*env)->GetLongArrayElements(env, offs, NULL) : NULL;
    jbyte *nr = o ? (*env)->GetByteArrayElements(env, nRows, NULL) : NULL;
    jchar *b = nr ? (*env)->GetCharArrayElements(env, branch, NULL) : NULL;
    jfloat *sc = b ? (*env)->GetFloatArrayElements(env, score, NULL) : NULL;
    jdouble *w = sc ? (*env)->GetDoubleArrayElements(env, lwr, NULL) : NULL;
    jint *f = w ? (*env)->GetIntArrayElements(env, flags, NULL) : NULL;
    int rc = RK_ERR_NOMEM, bad_offsets = 0;
    if (f) {
        /* offsets must be monotone and stay inside seqs: the engine reads seqs[offs[i] .. offs[i+1]) */
        bad_offsets = o[0] < 0 || o[n] > (jlong)nSeq;
        for (jsize i = 0; i < n && !bad_offsets; i++) bad_offsets = o[i + 1] < o[i];
        if (!bad_offsets) {
            rk_result out = {(uint8_t *)nr, (uint16_t *)b, sc, w, (uint32_t *)f};
            rc = n_handles == 1 ? rk_place_batch(handles[0], &p, (uint64_t)n, (const uint8_t *)s, (const uint64_t *)o, &out, NULL)
                                : rk_place_batch_multi(handles, n_handles, &p, (uint64_t)n, (const uint8_t *)s, (const uint64_t *)o, &out, NULL);
        }
    }
    const jint mode = (f && !bad_offsets && rc == RK_OK) ? 0 : JNI_ABORT;  /* copy results back only when there are results */
    if (f) (*env)->ReleaseIntArrayElements(env, flags, f, mode);
    if (w) (*env)->ReleaseDoubleArrayElements(env, lwr, w, mode);
    if (sc) (*env)->ReleaseFloatArrayElements(env, score, sc, mode);
    if (b) (*env)->ReleaseCharArrayElements(env, branch, b, mode);
    if (nr) (*env)->ReleaseByteArrayElements(env, nRows, nr, mode);
    if (o) (*env)->ReleaseLongArrayElements(env, offs, o, JNI_ABORT);
    if (s) (*env)->ReleaseByteArrayElements(env, seqs, s, JNI_ABORT);
    if (!f) { if (!(*env)->ExceptionCheck(env)) THROW_OOM("placeBatch: could not pin the arrays"); return; }
    if (bad_offsets) { THROW_ARG("placeBatch: offs must be non-decreasing, start >= 0 and end <= seqs.length"); return; }
    if (rc != RK_OK) throw_rk(env, n_handles == 1 ? "rk_place_batch" : "rk_place_batch_multi");
}

/* void placeBatch(long db, byte[] seqs, long[] offs, int keepAtMost, float keepFactor, int ambMode, float nsBound,
 *                 byte[] nRows, char[] branch, float[] score, double[] lwr, int[] flags) */
JNIEXPORT void JNICALL Java_core_algos_NativePlacement_placeBatch(JNIEnv *env, jclass cls, jlong db, jbyteArray seqs,
        jlongArray offs, jint keepAtMost, jfloat keepFactor, jint ambMode, jfloat nsBound, jbyteArray nRows,
        jcharArray branch, jfloatArray score, jdoubleArray lwr, jintArray flags) {
    (void)cls;
    rk_db *h = (rk_db *)(intptr_t)db;
    if (!h) { THROW_ARG("placeBatch: null database handle"); return; }
    place_common(env, &h, 1, seqs, offs, keepAtMost, keepFactor, ambMode, nsBound, nRows, branch, score, lwr, flags);
}

JNIEXPORT void JNICALL Java_core_algos_NativePlacement_dbDestroy(JNIEnv *env, jclass cls, jlong db) {
    (void)env; (void)cls;
    rk_db_destroy((rk_db *)(intptr_t)db);
}

/* Object[] buildDb(int alphabet, int k, int nNodes, int nSites, int nStates, byte[] states, float[] ppLog10, char[] nodeBranch,
 *                  float thrLog10, boolean gapJumps, boolean limitTo1Jump, int[] gapOff, int[] gapLen, int device)
 * -> {long[] keyCodes, long[] rowOffsets, char[] branchIds, float[] scores}
 * The Java caller flattens arpr.getPProbas() for nodesTested ([node][site][rank]) and align.getGapIntervals() (CSR);
 * replaces the loops of src/main_v2/Main_DBBUILD_3.java:648-750. */
JNIEXPORT jobjectArray JNICALL Java_core_algos_NativePlacement_buildDb(JNIEnv *env, jclass cls, jint alphabet, jint k, jint nNodes,
        jint nSites, jint nStates, jbyteArray states, jfloatArray ppLog10, jcharArray nodeBranch, jfloat thrLog10,
        jboolean gapJumps, jboolean limitTo1Jump, jintArray gapOff, jintArray gapLen, jint device) {
    (void)cls;
    if (!states || !ppLog10 || !nodeBranch || (gapJumps && (!gapOff || !gapLen))) { THROW_ARG("buildDb: null array"); return NULL; }
    if (nNodes < 0 || nSites < 1 || nStates < 1 || k < 0) { THROW_ARG("buildDb: negative or zero dimension"); return NULL; }
    const jlong cells = (jlong)nNodes * (jlong)nSites * (jlong)nStates;
    if ((jlong)(*env)->GetArrayLength(env, states) != cells || (jlong)(*env)->GetArrayLength(env, ppLog10) != cells ||
        (*env)->GetArrayLength(env, nodeBranch) != nNodes) {
        THROW_ARG("buildDb: states / ppLog10 need nNodes * nSites * nStates elements, nodeBranch nNodes");
        return NULL;
    }
    if (gapJumps && (*env)->GetArrayLength(env, gapOff) != nSites + 1) { THROW_ARG("buildDb: gapOff needs nSites + 1 elements"); return NULL; }
    rk_build_desc d;
    rk_built_db b;
    memset(&d, 0, sizeof d);
    d.alphabet = (uint32_t)alphabet; d.k = (uint32_t)k; d.n_nodes = (uint32_t)nNodes; d.n_sites = (uint32_t)nSites;
    d.n_states = (uint32_t)nStates; d.do_gap_jumps = gapJumps ? 1u : 0u; d.limit_to_1_jump = limitTo1Jump ? 1u : 0u;
    d.thr_log10 = thrLog10; d.device = device;
    jbyte *st = (*env)->GetByteArrayElements(env, states, NULL);
    jfloat *pp = st ? (*env)->GetFloatArrayElements(env, ppLog10, NULL) : NULL;
    jchar *nb = pp ? (*env)->GetCharArrayElements(env, nodeBranch, NULL) : NULL;
    jint *go = (nb && gapJumps) ? (*env)->GetIntArrayElements(env, gapOff, NULL) : NULL;
    jint *gl = (go && gapJumps) ? (*env)->GetIntArrayElements(env, gapLen, NULL) : NULL;
    const int pinned = nb && (!gapJumps || (go && gl));
    int rc = RK_ERR_NOMEM, bad_gaps = 0;
    if (pinned) {
        if (gapJumps) bad_gaps = go[0] != 0 || go[nSites] < 0 || go[nSites] != (*env)->GetArrayLength(env, gapLen);
        if (!bad_gaps) {
            d.states = (const uint8_t *)st; d.pp_log10 = pp; d.node_branch = (const uint16_t *)nb;
            d.gap_off = (const uint32_t *)go; d.gap_len = (const int32_t *)gl;
            rc = rk_build_db(&d, &b);
        }
    }
    if (gl) (*env)->ReleaseIntArrayElements(env, gapLen, gl, JNI_ABORT);
    if (go) (*env)->ReleaseIntArrayElements(env, gapOff, go, JNI_ABORT);
    if (nb) (*env)->ReleaseCharArrayElements(env, nodeBranch, nb, JNI_ABORT);
    if (pp) (*env)->ReleaseFloatArrayElements(env, ppLog10, pp, JNI_ABORT);
    if (st) (*env)->ReleaseByteArrayElements(env, states, st, JNI_ABORT);
    if (!pinned) { if (!(*env)->ExceptionCheck(env)) THROW_OOM("buildDb: could not pin the arrays"); return NULL; }
    if (bad_gaps) { THROW_ARG("buildDb: gapOff must start at 0 and end at gapLen.length"); return NULL; }
    if (rc != RK_OK) { throw_rk(env, "rk_build_db"); return NULL; }
    jobjectArray res = NULL;
    if (b.n_entries > 0x7FFFFFF0ull || b.n_keys > 0x7FFFFFF0ull) {
        throw_new(env, "java/lang/RuntimeException", "rk_build_db: result does not fit Java arrays; build per node batch");
    } else {
        jlongArray kc = (*env)->NewLongArray(env, (jsize)b.n_keys);
        jlongArray ro = (*env)->NewLongArray(env, (jsize)b.n_keys + 1);
        jcharArray br = (*env)->NewCharArray(env, (jsize)b.n_entries);
        jfloatArray sc = (*env)->NewFloatArray(env, (jsize)b.n_entries);
        jclass obj = (*env)->FindClass(env, "java/lang/Object");
        if (kc && ro && br && sc && obj && (res = (*env)->NewObjectArray(env, 4, obj, NULL))) {
            (*env)->SetLongArrayRegion(env, kc, 0, (jsize)b.n_keys, (const jlong *)b.key_codes);
            (*env)->SetLongArrayRegion(env, ro, 0, (jsize)b.n_keys + 1, (const jlong *)b.row_offsets);
            (*env)->SetCharArrayRegion(env, br, 0, (jsize)b.n_entries, (const jchar *)b.branch_ids);
            (*env)->SetFloatArrayRegion(env, sc, 0, (jsize)b.n_entries, b.scores);
            (*env)->SetObjectArrayElement(env, res, 0, kc);
            (*env)->SetObjectArrayElement(env, res, 1, ro);
            (*env)->SetObjectArrayElement(env, res, 2, br);
            (*env)->SetObjectArrayElement(env, res, 3, sc);
        }
    }
    rk_built_free(&b);
    return res;
}

/* void placeBatchMulti(long[] dbs, byte[] seqs, long[] offs, int keepAtMost, float keepFactor, int ambMode, float nsBound,
 *                      byte[] nRows, char[] branch, float[] score, double[] lwr, int[] flags)
 * dbs[g] = dbCreate(..., device = g): one handle per GPU, contiguous shards on host threads inside the library. */
JNIEXPORT void JNICALL Java_core_algos_NativePlacement_placeBatchMulti(JNIEnv *env, jclass cls, jlongArray dbs, jbyteArray seqs,
        jlongArray offs, jint keepAtMost, jfloat keepFactor, jint ambMode, jfloat nsBound, jbyteArray nRows,
        jcharArray branch, jfloatArray score, jdoubleArray lwr, jintArray flags) {
    (void)cls;
    if (!dbs) { THROW_ARG("placeBatchMulti: null handle array"); return; }
    const jsize nd = (*env)->GetArrayLength(env, dbs);
    rk_db *handles[64];
    if (nd < 1 || nd > 64) { THROW_ARG("placeBatchMulti: 1..64 database handles"); return; }
    jlong *h = (*env)->GetLongArrayElements(env, dbs, NULL);
    if (!h) { if (!(*env)->ExceptionCheck(env)) THROW_OOM("placeBatchMulti: could not pin the handle array"); return; }
    int null_handle = 0;
    for (jsize g = 0; g < nd; g++) { handles[g] = (rk_db *)(intptr_t)h[g]; null_handle |= handles[g] == NULL; }
    (*env)->ReleaseLongArrayElements(env, dbs, h, JNI_ABORT);
    if (null_handle) { THROW_ARG("placeBatchMulti: null database handle"); return; }
    place_common(env, handles, (uint32_t)nd, seqs, offs, keepAtMost, keepFactor, ambMode, nsBound, nRows, branch, score, lwr, flags);
}

/* ---- round 4: the HBM image as a file (rk_db_save / rk_db_load / rk_db_image_user) and the first call's set-up ahead of time ----
 * void dbSave(long db, String path, byte[] user) */
JNIEXPORT void JNICALL Java_core_algos_NativePlacement_dbSave(JNIEnv *env, jclass cls, jlong db, jstring path, jbyteArray user) {
    (void)cls;
    if (!db || !path) { THROW_ARG("dbSave: null handle or path"); return; }
    const char *p = (*env)->GetStringUTFChars(env, path, NULL);
    if (!p) return;
    jsize ulen = user ? (*env)->GetArrayLength(env, user) : 0;
    jbyte *u = ulen ? (*env)->GetByteArrayElements(env, user, NULL) : NULL;
    if (ulen && !u) { (*env)->ReleaseStringUTFChars(env, path, p); return; }
    const int rc = rk_db_save((rk_db *)(intptr_t)db, p, u, (uint64_t)ulen);
    if (u) (*env)->ReleaseByteArrayElements(env, user, u, JNI_ABORT);
    (*env)->ReleaseStringUTFChars(env, path, p);
    if (rc != RK_OK) throw_rk(env, "rk_db_save");
}

/* long dbLoad(String path, int device) */
JNIEXPORT jlong JNICALL Java_core_algos_NativePlacement_dbLoad(JNIEnv *env, jclass cls, jstring path, jint device) {
    (void)cls;
    if (!path) { THROW_ARG("dbLoad: null path"); return 0; }
    const char *p = (*env)->GetStringUTFChars(env, path, NULL);
    if (!p) return 0;
    rk_db *db = NULL;
    const int rc = rk_db_load(p, (int)device, &db);
    (*env)->ReleaseStringUTFChars(env, path, p);
    if (rc != RK_OK) { throw_rk(env, "rk_db_load"); return 0; }
    return (jlong)(intptr_t)db;
}

/* byte[] dbImageUser(String path): the caller's blob of an image file (the drivers keep the reference tree there) */
JNIEXPORT jbyteArray JNICALL Java_core_algos_NativePlacement_dbImageUser(JNIEnv *env, jclass cls, jstring path) {
    (void)cls;
    if (!path) { THROW_ARG("dbImageUser: null path"); return NULL; }
    const char *p = (*env)->GetStringUTFChars(env, path, NULL);
    if (!p) return NULL;
    uint64_t len = 0;
    jbyteArray out = NULL;
    if (rk_db_image_user(p, NULL, 0, &len) != RK_OK) {
        throw_rk(env, "rk_db_image_user");
    } else if (len > 0x7FFFFFF0ull) {
        THROW_ARG("dbImageUser: blob larger than a Java array");
    } else if ((out = (*env)->NewByteArray(env, (jsize)len)) != NULL && len) {
        jbyte *b = (*env)->GetByteArrayElements(env, out, NULL);
        if (b) {
            const int rc = rk_db_image_user(p, b, len, &len);
            (*env)->ReleaseByteArrayElements(env, out, b, 0);
            if (rc != RK_OK) { throw_rk(env, "rk_db_image_user"); out = NULL; }
        } else {
            out = NULL;
        }
    }
    (*env)->ReleaseStringUTFChars(env, path, p);
    return out;
}

/* void reserveHostPath(long db, int keepAtMost, int maxReadLen) */
JNIEXPORT void JNICALL Java_core_algos_NativePlacement_reserveHostPath(JNIEnv *env, jclass cls, jlong db, jint keepAtMost, jint maxReadLen) {
    (void)cls;
    if (!db || keepAtMost < 1 || maxReadLen < 1) { THROW_ARG("reserveHostPath: bad argument"); return; }
    if (rk_reserve_host_path((rk_db *)(intptr_t)db, (uint32_t)keepAtMost, (uint32_t)maxReadLen) != RK_OK) throw_rk(env, "rk_reserve_host_path");
}
