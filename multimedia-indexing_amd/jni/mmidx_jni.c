/*
 * mmidx_jni.c -- thin JNI shim between the reference's Java classes and the C ABI of
 * include/mmidx.h (libmmidx_hip.so).  No arithmetic happens here: arrays are pinned / copied,
 * status codes are turned back into the reference's `throw new Exception(msg)`.
 *
 * NOT compiled in the build container (no JDK, no jni.h).  On a box with a JDK:
 *   gcc -shared -fPIC -I$JAVA_HOME/include -I$JAVA_HOME/include/linux -I../../include \
 *       mmidx_jni.c -o libmmidx_jni.so -L../csrc -lmmidx_hip
 * Java side: java/gr/iti/mklab/visual/datastructures/{MmidxNative,GpuIVFPQ,GpuPQ,GpuLinear}.java
 */
#include <jni.h>
#include <stdint.h>
#include <stdlib.h>

#include "mmidx.h"

static void throw_status(JNIEnv *env, int status) {
    /* every reference error on this path is a checked java.lang.Exception with a message
     * (ASS:283, IVFPQ.java:182, :311, :359); capacity / duplicate ids never reach native code */
    jclass cls = (*env)->FindClass(env, status == MMIDX_ERR_INVALID_ARG ? "java/lang/IllegalArgumentException"
                                                                         : "java/lang/Exception");
    if (cls) (*env)->ThrowNew(env, cls, mmidx_last_error());
}
#define H(handle) ((mmidx_index *)(intptr_t)(handle))
#define CHECK(expr)                      \
    do {                                 \
        int st__ = (expr);               \
        if (st__ != MMIDX_OK) {          \
            throw_status(env, st__);     \
            goto done;                   \
        }                                \
    } while (0)

JNIEXPORT jlong JNICALL Java_gr_iti_mklab_visual_datastructures_MmidxNative_create(
    JNIEnv *env, jclass c, jint kind, jint D, jint m, jint ks, jint C, jint transform, jintArray perm,
    jdoubleArray rot, jint device) {
    mmidx_index *h = NULL;
    jint *p = perm ? (*env)->GetIntArrayElements(env, perm, NULL) : NULL;
    jdouble *r = rot ? (*env)->GetDoubleArrayElements(env, rot, NULL) : NULL;
    (void)c;
    CHECK(mmidx_create(kind, D, m, ks, C, transform, (const int32_t *)p, r, device, &h));
done:
    if (p) (*env)->ReleaseIntArrayElements(env, perm, p, JNI_ABORT);
    if (r) (*env)->ReleaseDoubleArrayElements(env, rot, r, JNI_ABORT);
    return (jlong)(intptr_t)h;
}

JNIEXPORT void JNICALL Java_gr_iti_mklab_visual_datastructures_MmidxNative_destroy(JNIEnv *env, jclass c, jlong h) {
    (void)env;
    (void)c;
    mmidx_destroy(H(h));
}

JNIEXPORT void JNICALL Java_gr_iti_mklab_visual_datastructures_MmidxNative_setCoarse(JNIEnv *env, jclass c, jlong h,
                                                                                      jdoubleArray flat) {
    jdouble *a = (*env)->GetDoubleArrayElements(env, flat, NULL);
    (void)c;
    CHECK(mmidx_set_coarse(H(h), a));
done:
    (*env)->ReleaseDoubleArrayElements(env, flat, a, JNI_ABORT);
}

JNIEXPORT void JNICALL Java_gr_iti_mklab_visual_datastructures_MmidxNative_setPq(JNIEnv *env, jclass c, jlong h,
                                                                                  jdoubleArray flat) {
    jdouble *a = (*env)->GetDoubleArrayElements(env, flat, NULL);
    (void)c;
    CHECK(mmidx_set_pq(H(h), a));
done:
    (*env)->ReleaseDoubleArrayElements(env, flat, a, JNI_ABORT);
}

JNIEXPORT void JNICALL Java_gr_iti_mklab_visual_datastructures_MmidxNative_setW(JNIEnv *env, jclass c, jlong h, jint w) {
    (void)c;
    CHECK(mmidx_set_w(H(h), w));
done:
    return;
}

/* indexVectorInternal: encode + append one vector; returns {cell, code bytes...} so that the Java
 * side can run appendPersistentIndex (IVFPQ.java:760-772) unchanged.  out = byte[4 + m]:
 * big-endian cell followed by the m stored bytes. */
JNIEXPORT void JNICALL Java_gr_iti_mklab_visual_datastructures_MmidxNative_addVector(
    JNIEnv *env, jclass c, jlong h, jint iid, jdoubleArray vec, jintArray cellOut, jbyteArray codeOut) {
    jdouble *v = (*env)->GetDoubleArrayElements(env, vec, NULL);
    jbyte *code = (*env)->GetByteArrayElements(env, codeOut, NULL);
    int32_t cell = -1, id = iid;
    (void)c;
    CHECK(mmidx_add_vectors(H(h), 1, v, &id, &cell, code));
    (*env)->SetIntArrayRegion(env, cellOut, 0, 1, (const jint *)&cell);
done:
    (*env)->ReleaseDoubleArrayElements(env, vec, v, JNI_ABORT);
    (*env)->ReleaseByteArrayElements(env, codeOut, code, 0);
}

/* indexPQCode / loadIndexInMemory: append n precomputed records */
JNIEXPORT void JNICALL Java_gr_iti_mklab_visual_datastructures_MmidxNative_addCodes(
    JNIEnv *env, jclass c, jlong h, jint n, jintArray iids, jintArray cells, jbyteArray codes) {
    jint *i = (*env)->GetIntArrayElements(env, iids, NULL);
    jint *l = cells ? (*env)->GetIntArrayElements(env, cells, NULL) : NULL;
    jbyte *k = (*env)->GetByteArrayElements(env, codes, NULL);
    (void)c;
    CHECK(mmidx_add_codes(H(h), n, (const int32_t *)i, (const int32_t *)l, k));
done:
    (*env)->ReleaseIntArrayElements(env, iids, i, JNI_ABORT);
    if (l) (*env)->ReleaseIntArrayElements(env, cells, l, JNI_ABORT);
    (*env)->ReleaseByteArrayElements(env, codes, k, JNI_ABORT);
}

/* computeNearestNeighborsInternal for nq queries (nq = 1 for the reference's single-query call).
 * Returns the per-query counts; iids / dists are filled row-major [nq][k]. */
JNIEXPORT void JNICALL Java_gr_iti_mklab_visual_datastructures_MmidxNative_search(
    JNIEnv *env, jclass c, jlong h, jint k, jint nq, jdoubleArray queries, jintArray iidOut, jdoubleArray distOut,
    jintArray countOut) {
    jdouble *q = (*env)->GetDoubleArrayElements(env, queries, NULL);
    jint *ii = (*env)->GetIntArrayElements(env, iidOut, NULL);
    jdouble *dd = (*env)->GetDoubleArrayElements(env, distOut, NULL);
    jint *cc = (*env)->GetIntArrayElements(env, countOut, NULL);
    (void)c;
    CHECK(mmidx_search(H(h), k, nq, q, (int32_t *)ii, dd, (int32_t *)cc));
done:
    (*env)->ReleaseDoubleArrayElements(env, queries, q, JNI_ABORT);
    (*env)->ReleaseIntArrayElements(env, iidOut, ii, 0);
    (*env)->ReleaseDoubleArrayElements(env, distOut, dd, 0);
    (*env)->ReleaseIntArrayElements(env, countOut, cc, 0);
}

JNIEXPORT void JNICALL Java_gr_iti_mklab_visual_datastructures_MmidxNative_searchSdc(
    JNIEnv *env, jclass c, jlong h, jint k, jint nq, jintArray queryIids, jintArray iidOut, jdoubleArray distOut,
    jintArray countOut) {
    jint *q = (*env)->GetIntArrayElements(env, queryIids, NULL);
    jint *ii = (*env)->GetIntArrayElements(env, iidOut, NULL);
    jdouble *dd = (*env)->GetDoubleArrayElements(env, distOut, NULL);
    jint *cc = (*env)->GetIntArrayElements(env, countOut, NULL);
    (void)c;
    CHECK(mmidx_search_sdc(H(h), k, nq, (const int32_t *)q, (int32_t *)ii, dd, (int32_t *)cc));
done:
    (*env)->ReleaseIntArrayElements(env, queryIids, q, JNI_ABORT);
    (*env)->ReleaseIntArrayElements(env, iidOut, ii, 0);
    (*env)->ReleaseDoubleArrayElements(env, distOut, dd, 0);
    (*env)->ReleaseIntArrayElements(env, countOut, cc, 0);
}

JNIEXPORT void JNICALL Java_gr_iti_mklab_visual_datastructures_MmidxNative_listSizes(JNIEnv *env, jclass c, jlong h,
                                                                                      jintArray out) {
    jint *o = (*env)->GetIntArrayElements(env, out, NULL);
    (void)c;
    CHECK(mmidx_list_sizes(H(h), (int32_t *)o));
done:
    (*env)->ReleaseIntArrayElements(env, out, o, 0);
}

/* ---- Linear ---------------------------------------------------------------------------------- */
#define HL(handle) ((mmidx_linear *)(intptr_t)(handle))

JNIEXPORT jlong JNICALL Java_gr_iti_mklab_visual_datastructures_MmidxNative_linearCreate(JNIEnv *env, jclass c, jint D,
                                                                                          jlong cap, jint device) {
    mmidx_linear *l = NULL;
    (void)c;
    CHECK(mmidx_linear_create(D, cap, device, &l));
done:
    return (jlong)(intptr_t)l;
}

JNIEXPORT void JNICALL Java_gr_iti_mklab_visual_datastructures_MmidxNative_linearDestroy(JNIEnv *env, jclass c, jlong h) {
    (void)env;
    (void)c;
    mmidx_linear_destroy(HL(h));
}

JNIEXPORT void JNICALL Java_gr_iti_mklab_visual_datastructures_MmidxNative_linearAdd(JNIEnv *env, jclass c, jlong h, jint n,
                                                                                      jdoubleArray flat) {
    jdouble *a = (*env)->GetDoubleArrayElements(env, flat, NULL);
    (void)c;
    CHECK(mmidx_linear_add(HL(h), n, a));
done:
    (*env)->ReleaseDoubleArrayElements(env, flat, a, JNI_ABORT);
}

JNIEXPORT void JNICALL Java_gr_iti_mklab_visual_datastructures_MmidxNative_linearSearch(
    JNIEnv *env, jclass c, jlong h, jint k, jint nq, jdoubleArray queries, jintArray iidOut, jdoubleArray distOut,
    jintArray countOut) {
    jdouble *q = (*env)->GetDoubleArrayElements(env, queries, NULL);
    jint *ii = (*env)->GetIntArrayElements(env, iidOut, NULL);
    jdouble *dd = (*env)->GetDoubleArrayElements(env, distOut, NULL);
    jint *cc = (*env)->GetIntArrayElements(env, countOut, NULL);
    (void)c;
    CHECK(mmidx_linear_search(HL(h), k, nq, q, (int32_t *)ii, dd, (int32_t *)cc));
done:
    (*env)->ReleaseDoubleArrayElements(env, queries, q, JNI_ABORT);
    (*env)->ReleaseIntArrayElements(env, iidOut, ii, 0);
    (*env)->ReleaseDoubleArrayElements(env, distOut, dd, 0);
    (*env)->ReleaseIntArrayElements(env, countOut, cc, 0);
}
