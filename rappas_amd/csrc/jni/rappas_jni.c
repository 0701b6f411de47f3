/*
 * rappas_jni.c -- thin JNI adapter over the C ABI of include/rappas_place.h.
 *
 * Java side: class core.algos.NativePlacement.  This file is compiled only when a JDK is
 * available (JAVA_HOME set): this image has no JVM and no jni.h, so it is written blind and is NOT part of
 * build()/tests.  Build:
 *   gcc -shared -fPIC -I$JAVA_HOME/include -I$JAVA_HOME/include/linux -Iinclude \
 *       rappas_amd/csrc/jni/rappas_jni.c -Lrappas_amd -lrappas_place -o librappas_jni.so
 */
#include <jni.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "rappas_place.h"

static void throw_rk(JNIEnv *env, const char *where) {
    char msg[640];
    snprintf(msg, sizeof msg, "%s: %s", where, rk_last_error());
    jclass ex = (*env)->FindClass(env, "java/lang/RuntimeException");
    if (ex) (*env)->ThrowNew(env, ex, msg);
}

/* long dbCreate(int alphabet, boolean convertUO, int k, int nBranches, float thrLog10, float thr,
 *               long[] keyCodes, long[] rowOffsets, char[] branchIds, float[] scores, int device)
 * The Java caller fills the arrays by walking session.hash exactly like SessionNext_v2.saveToJSON
 * (src/main_v2/SessionNext_v2.java:250-261). */
JNIEXPORT jlong JNICALL Java_core_algos_NativePlacement_dbCreate(JNIEnv *env, jclass cls, jint alphabet,
        jboolean convertUO, jint k, jint nBranches, jfloat thrLog10, jfloat thr, jlongArray keyCodes,
        jlongArray rowOffsets, jcharArray branchIds, jfloatArray scores, jint device) {
    (void)cls;
    rk_db_desc d;
    d.alphabet = (uint32_t)alphabet; d.convert_uo = convertUO ? 1u : 0u; d.k = (uint32_t)k;
    d.n_branches = (uint32_t)nBranches; d.thr_log10 = thrLog10; d.thr = thr;
    d.n_keys = (uint64_t)(*env)->GetArrayLength(env, keyCodes);
    d.device = device; d.table_mode = RK_TABLE_AUTO;
    jlong *kc = (*env)->GetLongArrayElements(env, keyCodes, NULL);
    jlong *ro = (*env)->GetLongArrayElements(env, rowOffsets, NULL);
    jchar *br = (*env)->GetCharArrayElements(env, branchIds, NULL);   /* jchar == uint16: (char)nodeId */
    jfloat *sc = (*env)->GetFloatArrayElements(env, scores, NULL);
    d.key_codes = (const uint64_t *)kc; d.row_offsets = (const uint64_t *)ro;
    d.branch_ids = (const uint16_t *)br; d.scores = sc;
    rk_db *db = NULL;
    int rc = rk_db_create(&d, &db);
    (*env)->ReleaseLongArrayElements(env, keyCodes, kc, JNI_ABORT);
    (*env)->ReleaseLongArrayElements(env, rowOffsets, ro, JNI_ABORT);
    (*env)->ReleaseCharArrayElements(env, branchIds, br, JNI_ABORT);
    (*env)->ReleaseFloatArrayElements(env, scores, sc, JNI_ABORT);
    if (rc != RK_OK) { throw_rk(env, "rk_db_create"); return 0; }
    return (jlong)(intptr_t)db;
}

/* void placeBatch(long db, byte[] seqs, long[] offs, int keepAtMost, float keepFactor, int ambMode, float nsBound,
 *                 byte[] nRows, char[] branch, float[] score, double[] lwr, int[] flags) */
JNIEXPORT void JNICALL Java_core_algos_NativePlacement_placeBatch(JNIEnv *env, jclass cls, jlong db, jbyteArray seqs,
        jlongArray offs, jint keepAtMost, jfloat keepFactor, jint ambMode, jfloat nsBound, jbyteArray nRows,
        jcharArray branch, jfloatArray score, jdoubleArray lwr, jintArray flags) {
    (void)cls;
    rk_params p = {(uint32_t)keepAtMost, keepFactor, (uint32_t)ambMode, nsBound};
    const uint64_t n = (uint64_t)(*env)->GetArrayLength(env, offs) - 1;
    jbyte *s = (*env)->GetByteArrayElements(env, seqs, NULL);
    jlong *o = (*env)->GetLongArrayElements(env, offs, NULL);
    jbyte *nr = (*env)->GetByteArrayElements(env, nRows, NULL);
    jchar *b = (*env)->GetCharArrayElements(env, branch, NULL);
    jfloat *sc = (*env)->GetFloatArrayElements(env, score, NULL);
    jdouble *w = (*env)->GetDoubleArrayElements(env, lwr, NULL);
    jint *f = (*env)->GetIntArrayElements(env, flags, NULL);
    rk_result out = {(uint8_t *)nr, (uint16_t *)b, sc, w, (uint32_t *)f};
    int rc = rk_place_batch((rk_db *)(intptr_t)db, &p, n, (const uint8_t *)s, (const uint64_t *)o, &out, NULL);
    (*env)->ReleaseByteArrayElements(env, seqs, s, JNI_ABORT);
    (*env)->ReleaseLongArrayElements(env, offs, o, JNI_ABORT);
    (*env)->ReleaseByteArrayElements(env, nRows, nr, 0);
    (*env)->ReleaseCharArrayElements(env, branch, b, 0);
    (*env)->ReleaseFloatArrayElements(env, score, sc, 0);
    (*env)->ReleaseDoubleArrayElements(env, lwr, w, 0);
    (*env)->ReleaseIntArrayElements(env, flags, f, 0);
    if (rc != RK_OK) throw_rk(env, "rk_place_batch");
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
    rk_build_desc d;
    rk_built_db b;
    memset(&d, 0, sizeof d);
    d.alphabet = (uint32_t)alphabet; d.k = (uint32_t)k; d.n_nodes = (uint32_t)nNodes; d.n_sites = (uint32_t)nSites;
    d.n_states = (uint32_t)nStates; d.do_gap_jumps = gapJumps ? 1u : 0u; d.limit_to_1_jump = limitTo1Jump ? 1u : 0u;
    d.thr_log10 = thrLog10; d.device = device;
    jbyte *st = (*env)->GetByteArrayElements(env, states, NULL);
    jfloat *pp = (*env)->GetFloatArrayElements(env, ppLog10, NULL);
    jchar *nb = (*env)->GetCharArrayElements(env, nodeBranch, NULL);
    jint *go = gapJumps ? (*env)->GetIntArrayElements(env, gapOff, NULL) : NULL;
    jint *gl = gapJumps ? (*env)->GetIntArrayElements(env, gapLen, NULL) : NULL;
    d.states = (const uint8_t *)st; d.pp_log10 = pp; d.node_branch = (const uint16_t *)nb;
    d.gap_off = (const uint32_t *)go; d.gap_len = (const int32_t *)gl;
    int rc = rk_build_db(&d, &b);
    (*env)->ReleaseByteArrayElements(env, states, st, JNI_ABORT);
    (*env)->ReleaseFloatArrayElements(env, ppLog10, pp, JNI_ABORT);
    (*env)->ReleaseCharArrayElements(env, nodeBranch, nb, JNI_ABORT);
    if (go) (*env)->ReleaseIntArrayElements(env, gapOff, go, JNI_ABORT);
    if (gl) (*env)->ReleaseIntArrayElements(env, gapLen, gl, JNI_ABORT);
    if (rc != RK_OK) { throw_rk(env, "rk_build_db"); return NULL; }
    jobjectArray res = NULL;
    if (b.n_entries > 0x7FFFFFF0ull || b.n_keys > 0x7FFFFFF0ull) {
        jclass ex = (*env)->FindClass(env, "java/lang/RuntimeException");
        if (ex) (*env)->ThrowNew(env, ex, "rk_build_db: result does not fit Java arrays; build per node batch");
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
    rk_params p = {(uint32_t)keepAtMost, keepFactor, (uint32_t)ambMode, nsBound};
    const uint64_t n = (uint64_t)(*env)->GetArrayLength(env, offs) - 1;
    const jsize nd = (*env)->GetArrayLength(env, dbs);
    rk_db *handles[64];
    if (nd < 1 || nd > 64) {
        jclass ex = (*env)->FindClass(env, "java/lang/IllegalArgumentException");
        if (ex) (*env)->ThrowNew(env, ex, "placeBatchMulti: 1..64 database handles");
        return;
    }
    jlong *h = (*env)->GetLongArrayElements(env, dbs, NULL);
    for (jsize g = 0; g < nd; g++) handles[g] = (rk_db *)(intptr_t)h[g];
    (*env)->ReleaseLongArrayElements(env, dbs, h, JNI_ABORT);
    jbyte *s = (*env)->GetByteArrayElements(env, seqs, NULL);
    jlong *o = (*env)->GetLongArrayElements(env, offs, NULL);
    jbyte *nr = (*env)->GetByteArrayElements(env, nRows, NULL);
    jchar *b = (*env)->GetCharArrayElements(env, branch, NULL);
    jfloat *sc = (*env)->GetFloatArrayElements(env, score, NULL);
    jdouble *w = (*env)->GetDoubleArrayElements(env, lwr, NULL);
    jint *f = (*env)->GetIntArrayElements(env, flags, NULL);
    rk_result out = {(uint8_t *)nr, (uint16_t *)b, sc, w, (uint32_t *)f};
    int rc = rk_place_batch_multi(handles, (uint32_t)nd, &p, n, (const uint8_t *)s, (const uint64_t *)o, &out, NULL);
    (*env)->ReleaseByteArrayElements(env, seqs, s, JNI_ABORT);
    (*env)->ReleaseLongArrayElements(env, offs, o, JNI_ABORT);
    (*env)->ReleaseByteArrayElements(env, nRows, nr, 0);
    (*env)->ReleaseCharArrayElements(env, branch, b, 0);
    (*env)->ReleaseFloatArrayElements(env, score, sc, 0);
    (*env)->ReleaseDoubleArrayElements(env, lwr, w, 0);
    (*env)->ReleaseIntArrayElements(env, flags, f, 0);
    if (rc != RK_OK) throw_rk(env, "rk_place_batch_multi");
}
