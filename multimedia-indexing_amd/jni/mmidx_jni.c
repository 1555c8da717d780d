/*
 * mmidx_jni.c -- thin JNI shim between the reference's Java classes and the C ABI of
 * include/mmidx.h (libmmidx_hip.so).  No arithmetic happens here: arrays are pinned / copied,
 * status codes are turned back into the reference's `throw new Exception(msg)`.
 *
 * Every entry point checks the length of every Java array against what the C call will read or write
 * (mmidx_get_dims supplies D, m, ks, C and the bytes per code entry) BEFORE pinning anything: a mismatch
 * throws IllegalArgumentException instead of touching memory past the array.  Codes cross as byte[] when
 * numProductCentroids <= 256 and as short[] otherwise, as in the reference (IVFPQ.java:342-354, PQ.java:544-558);
 * calling the wrong variant throws.
 *
 * NOT compiled in the build container (no JDK, no jni.h).  On a box with a JDK: CMakeLists.txt next to this
 * file (find_package(JNI)), or
 *   gcc -shared -fPIC -I$JAVA_HOME/include -I$JAVA_HOME/include/linux -I../../include \
 *       mmidx_jni.c -o libmmidx_jni.so -L../csrc -lmmidx_hip
 * Java side: java/gr/iti/mklab/visual/datastructures/{MmidxNative,GpuIVFPQ,GpuPQ,GpuLinear}.java,
 *            java/gr/iti/mklab/visual/dimreduction/GpuPCA.java, java/gr/iti/mklab/visual/aggregation/GpuVladAggregator.java
 */
#include <jni.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "mmidx.h"

#define JFN(name) Java_gr_iti_mklab_visual_datastructures_MmidxNative_##name

static void throw_msg(JNIEnv *env, const char *cls_name, const char *msg) {
    jclass cls = (*env)->FindClass(env, cls_name);
    if (cls) (*env)->ThrowNew(env, cls, msg);
}
static void throw_status(JNIEnv *env, int status) {
    /* every reference error on this path is a checked java.lang.Exception with a message
     * (ASS:283, IVFPQ.java:182, :311, :359, :805); capacity / duplicate ids never reach native code */
    (void)status;
    throw_msg(env, "java/lang/Exception", mmidx_last_error());
}
/* 0 when array `a` (may be NULL iff allow_null) holds at least `need` elements; else throws and returns 1 */
static int bad_len(JNIEnv *env, jarray a, int64_t need, int allow_null, const char *what) {
    char buf[160];
    if (!a) {
        if (allow_null) return 0;
        snprintf(buf, sizeof(buf), "%s is null", what);
        throw_msg(env, "java/lang/IllegalArgumentException", buf);
        return 1;
    }
    if (need < 0 || (int64_t)(*env)->GetArrayLength(env, a) < need) {
        snprintf(buf, sizeof(buf), "%s has %d elements, the native call needs %lld", what, (int)(*env)->GetArrayLength(env, a),
                 (long long)need);
        throw_msg(env, "java/lang/IllegalArgumentException", buf);
        return 1;
    }
    return 0;
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

typedef struct {
    int D, m, ks, C, cb;
} dims_t;
static int get_dims(JNIEnv *env, jlong h, dims_t *d) {
    if (mmidx_get_dims(H(h), &d->D, &d->m, &d->ks, &d->C, &d->cb) != MMIDX_OK) {
        throw_status(env, MMIDX_ERR_INVALID_ARG);
        return 1;
    }
    return 0;
}
/* the code variant (1 = byte[], 2 = short[]) must match the handle */
static int bad_variant(JNIEnv *env, const dims_t *d, int elem_bytes) {
    if (d->cb == elem_bytes) return 0;
    throw_msg(env, "java/lang/Exception",
              elem_bytes == 1 ? "Byte is not sufficient to enumerate the centroids of the product quantizer!" /* IVFPQ.java:358-361 */
                              : "short[] codes need numProductCentroids > 256");
    return 1;
}

JNIEXPORT jlong JNICALL JFN(create)(JNIEnv *env, jclass c, jint kind, jint D, jint m, jint ks, jint C, jint transform, jintArray perm,
                                    jdoubleArray rot, jint device) {
    mmidx_index *h = NULL;
    jint *p = NULL;
    jdouble *r = NULL;
    (void)c;
    if (bad_len(env, perm, D, 1, "permutation") || bad_len(env, rot, (int64_t)D * D, 1, "rotation")) return 0;
    p = perm ? (*env)->GetIntArrayElements(env, perm, NULL) : NULL;
    r = rot ? (*env)->GetDoubleArrayElements(env, rot, NULL) : NULL;
    CHECK(mmidx_create(kind, D, m, ks, C, transform, (const int32_t *)p, r, device, &h));
done:
    if (p) (*env)->ReleaseIntArrayElements(env, perm, p, JNI_ABORT);
    if (r) (*env)->ReleaseDoubleArrayElements(env, rot, r, JNI_ABORT);
    return (jlong)(intptr_t)h;
}

/* mmidx_create_sharded: ONE index over several GPUs of the node, driven from this JVM (the reference's caller is one process
 * holding the whole index, YFCC100MExample.java:93-99); every other entry point takes the returned handle unchanged */
JNIEXPORT jlong JNICALL JFN(createSharded)(JNIEnv *env, jclass c, jint kind, jint D, jint m, jint ks, jint C, jint transform, jintArray perm,
                                           jdoubleArray rot, jintArray devices) {
    mmidx_index *h = NULL;
    jint *p = NULL, *dv = NULL;
    jdouble *r = NULL;
    jint ndev;
    (void)c;
    if (bad_len(env, perm, D, 1, "permutation") || bad_len(env, rot, (int64_t)D * D, 1, "rotation") || bad_len(env, devices, 1, 0, "devices")) return 0;
    ndev = (*env)->GetArrayLength(env, devices);
    p = perm ? (*env)->GetIntArrayElements(env, perm, NULL) : NULL;
    r = rot ? (*env)->GetDoubleArrayElements(env, rot, NULL) : NULL;
    dv = (*env)->GetIntArrayElements(env, devices, NULL);
    CHECK(mmidx_create_sharded(kind, D, m, ks, C, transform, (const int32_t *)p, r, ndev, (const int *)dv, &h));
done:
    if (p) (*env)->ReleaseIntArrayElements(env, perm, p, JNI_ABORT);
    if (r) (*env)->ReleaseDoubleArrayElements(env, rot, r, JNI_ABORT);
    if (dv) (*env)->ReleaseIntArrayElements(env, devices, dv, JNI_ABORT);
    return (jlong)(intptr_t)h;
}

JNIEXPORT void JNICALL JFN(destroy)(JNIEnv *env, jclass c, jlong h) {
    (void)env;
    (void)c;
    mmidx_destroy(H(h));
}

JNIEXPORT void JNICALL JFN(setCoarse)(JNIEnv *env, jclass c, jlong h, jdoubleArray flat) {
    dims_t d;
    jdouble *a;
    (void)c;
    if (get_dims(env, h, &d) || bad_len(env, flat, (int64_t)d.C * d.D, 0, "coarse quantizer")) return;
    a = (*env)->GetDoubleArrayElements(env, flat, NULL);
    CHECK(mmidx_set_coarse(H(h), a));
done:
    (*env)->ReleaseDoubleArrayElements(env, flat, a, JNI_ABORT);
}

JNIEXPORT void JNICALL JFN(setPq)(JNIEnv *env, jclass c, jlong h, jdoubleArray flat) {
    dims_t d;
    jdouble *a;
    (void)c;
    if (get_dims(env, h, &d) || bad_len(env, flat, (int64_t)d.ks * d.D, 0, "product quantizer")) return; /* m * ks * (D / m) */
    a = (*env)->GetDoubleArrayElements(env, flat, NULL);
    CHECK(mmidx_set_pq(H(h), a));
done:
    (*env)->ReleaseDoubleArrayElements(env, flat, a, JNI_ABORT);
}

JNIEXPORT void JNICALL JFN(setW)(JNIEnv *env, jclass c, jlong h, jint w) {
    (void)c;
    CHECK(mmidx_set_w(H(h), w));
done:
    return;
}

/* indexVectorInternal: encode + append one vector; cellOut[0] / codeOut[m] receive the record so that the Java
 * side can run appendPersistentIndex (IVFPQ.java:760-792) unchanged.  Byte and short variants. */
static void add_vector(JNIEnv *env, jlong h, jint iid, jdoubleArray vec, jintArray cellOut, jarray codeOut, int elem_bytes) {
    dims_t d;
    jdouble *v;
    void *code;
    int32_t cell = -1, id = iid;
    if (get_dims(env, h, &d) || bad_variant(env, &d, elem_bytes) || bad_len(env, vec, d.D, 0, "vector") ||
        bad_len(env, cellOut, 1, 0, "cellOut") || bad_len(env, codeOut, d.m, 0, "codeOut"))
        return;
    v = (*env)->GetDoubleArrayElements(env, vec, NULL);
    code = elem_bytes == 1 ? (void *)(*env)->GetByteArrayElements(env, (jbyteArray)codeOut, NULL)
                           : (void *)(*env)->GetShortArrayElements(env, (jshortArray)codeOut, NULL);
    CHECK(mmidx_add_vectors(H(h), 1, v, &id, &cell, code));
    (*env)->SetIntArrayRegion(env, cellOut, 0, 1, (const jint *)&cell);
done:
    (*env)->ReleaseDoubleArrayElements(env, vec, v, JNI_ABORT);
    if (elem_bytes == 1) (*env)->ReleaseByteArrayElements(env, (jbyteArray)codeOut, (jbyte *)code, 0);
    else (*env)->ReleaseShortArrayElements(env, (jshortArray)codeOut, (jshort *)code, 0);
}
JNIEXPORT void JNICALL JFN(addVector)(JNIEnv *env, jclass c, jlong h, jint iid, jdoubleArray vec, jintArray cellOut, jbyteArray codeOut) {
    (void)c;
    add_vector(env, h, iid, vec, cellOut, codeOut, 1);
}
JNIEXPORT void JNICALL JFN(addVectorShort)(JNIEnv *env, jclass c, jlong h, jint iid, jdoubleArray vec, jintArray cellOut,
                                           jshortArray codeOut) {
    (void)c;
    add_vector(env, h, iid, vec, cellOut, codeOut, 2);
}

/* indexPQCode / loadIndexInMemory: append n precomputed records (cells == null for PQ) */
static void add_codes(JNIEnv *env, jlong h, jint n, jintArray iids, jintArray cells, jarray codes, int elem_bytes) {
    dims_t d;
    jint *i, *l;
    void *k;
    if (get_dims(env, h, &d) || bad_variant(env, &d, elem_bytes) || bad_len(env, iids, n, 0, "iids") || bad_len(env, cells, n, 1, "cells") ||
        bad_len(env, codes, (int64_t)n * d.m, 0, "codes"))
        return;
    i = (*env)->GetIntArrayElements(env, iids, NULL);
    l = cells ? (*env)->GetIntArrayElements(env, cells, NULL) : NULL;
    k = elem_bytes == 1 ? (void *)(*env)->GetByteArrayElements(env, (jbyteArray)codes, NULL)
                        : (void *)(*env)->GetShortArrayElements(env, (jshortArray)codes, NULL);
    CHECK(mmidx_add_codes(H(h), n, (const int32_t *)i, (const int32_t *)l, k));
done:
    (*env)->ReleaseIntArrayElements(env, iids, i, JNI_ABORT);
    if (l) (*env)->ReleaseIntArrayElements(env, cells, l, JNI_ABORT);
    if (elem_bytes == 1) (*env)->ReleaseByteArrayElements(env, (jbyteArray)codes, (jbyte *)k, JNI_ABORT);
    else (*env)->ReleaseShortArrayElements(env, (jshortArray)codes, (jshort *)k, JNI_ABORT);
}
JNIEXPORT void JNICALL JFN(addCodes)(JNIEnv *env, jclass c, jlong h, jint n, jintArray iids, jintArray cells, jbyteArray codes) {
    (void)c;
    add_codes(env, h, n, iids, cells, codes, 1);
}
JNIEXPORT void JNICALL JFN(addCodesShort)(JNIEnv *env, jclass c, jlong h, jint n, jintArray iids, jintArray cells, jshortArray codes) {
    (void)c;
    add_codes(env, h, n, iids, cells, codes, 2);
}

/* getInvertedListId / getPQCodeByte / getPQCodeShort for n internal ids (IVFPQ.java:801-880); cellsOut / codesOut may be null */
static void get_codes(JNIEnv *env, jlong h, jintArray iids, jintArray cellsOut, jarray codesOut, int elem_bytes) {
    dims_t d;
    jint n, *i, *l;
    void *k;
    if (bad_len(env, iids, 0, 0, "iids") || get_dims(env, h, &d)) return;
    n = (*env)->GetArrayLength(env, iids);
    if ((codesOut && bad_variant(env, &d, elem_bytes)) || bad_len(env, cellsOut, n, 1, "cellsOut") ||
        bad_len(env, codesOut, (int64_t)n * d.m, 1, "codesOut"))
        return;
    i = (*env)->GetIntArrayElements(env, iids, NULL);
    l = cellsOut ? (*env)->GetIntArrayElements(env, cellsOut, NULL) : NULL;
    k = !codesOut ? NULL
        : elem_bytes == 1 ? (void *)(*env)->GetByteArrayElements(env, (jbyteArray)codesOut, NULL)
                          : (void *)(*env)->GetShortArrayElements(env, (jshortArray)codesOut, NULL);
    CHECK(mmidx_get_codes(H(h), n, (const int32_t *)i, (int32_t *)l, k));
done:
    (*env)->ReleaseIntArrayElements(env, iids, i, JNI_ABORT);
    if (l) (*env)->ReleaseIntArrayElements(env, cellsOut, l, 0);
    if (k && elem_bytes == 1) (*env)->ReleaseByteArrayElements(env, (jbyteArray)codesOut, (jbyte *)k, 0);
    if (k && elem_bytes == 2) (*env)->ReleaseShortArrayElements(env, (jshortArray)codesOut, (jshort *)k, 0);
}
JNIEXPORT void JNICALL JFN(getCodes)(JNIEnv *env, jclass c, jlong h, jintArray iids, jintArray cellsOut, jbyteArray codesOut) {
    (void)c;
    get_codes(env, h, iids, cellsOut, codesOut, 1);
}
JNIEXPORT void JNICALL JFN(getCodesShort)(JNIEnv *env, jclass c, jlong h, jintArray iids, jintArray cellsOut, jshortArray codesOut) {
    (void)c;
    get_codes(env, h, iids, cellsOut, codesOut, 2);
}

/* computeDistanceIVFADC for n (query, internal id) pairs, IVFPQ.java:464-497 */
JNIEXPORT void JNICALL JFN(distance)(JNIEnv *env, jclass c, jlong h, jdoubleArray queries, jintArray iids, jdoubleArray out) {
    dims_t d;
    jint n, *i;
    jdouble *q, *o;
    (void)c;
    if (bad_len(env, iids, 0, 0, "iids") || get_dims(env, h, &d)) return;
    n = (*env)->GetArrayLength(env, iids);
    if (bad_len(env, queries, (int64_t)n * d.D, 0, "queries") || bad_len(env, out, n, 0, "out")) return;
    q = (*env)->GetDoubleArrayElements(env, queries, NULL);
    i = (*env)->GetIntArrayElements(env, iids, NULL);
    o = (*env)->GetDoubleArrayElements(env, out, NULL);
    CHECK(mmidx_distance(H(h), n, q, (const int32_t *)i, o));
done:
    (*env)->ReleaseDoubleArrayElements(env, queries, q, JNI_ABORT);
    (*env)->ReleaseIntArrayElements(env, iids, i, JNI_ABORT);
    (*env)->ReleaseDoubleArrayElements(env, out, o, 0);
}

/* computeNearestNeighborsInternal for nq queries (nq = 1 for the reference's single-query call).
 * Returns the per-query counts; iids / dists are filled row-major [nq][k]. */
JNIEXPORT void JNICALL JFN(search)(JNIEnv *env, jclass c, jlong h, jint k, jint nq, jdoubleArray queries, jintArray iidOut,
                                   jdoubleArray distOut, jintArray countOut) {
    dims_t d;
    jdouble *q, *dd;
    jint *ii, *cc;
    (void)c;
    if (get_dims(env, h, &d) || bad_len(env, queries, (int64_t)nq * d.D, 0, "queries") || bad_len(env, iidOut, (int64_t)nq * k, 0, "iidOut") ||
        bad_len(env, distOut, (int64_t)nq * k, 0, "distOut") || bad_len(env, countOut, nq, 0, "countOut"))
        return;
    q = (*env)->GetDoubleArrayElements(env, queries, NULL);
    ii = (*env)->GetIntArrayElements(env, iidOut, NULL);
    dd = (*env)->GetDoubleArrayElements(env, distOut, NULL);
    cc = (*env)->GetIntArrayElements(env, countOut, NULL);
    CHECK(mmidx_search(H(h), k, nq, q, (int32_t *)ii, dd, (int32_t *)cc));
done:
    (*env)->ReleaseDoubleArrayElements(env, queries, q, JNI_ABORT);
    (*env)->ReleaseIntArrayElements(env, iidOut, ii, 0);
    (*env)->ReleaseDoubleArrayElements(env, distOut, dd, 0);
    (*env)->ReleaseIntArrayElements(env, countOut, cc, 0);
}

JNIEXPORT void JNICALL JFN(searchSdc)(JNIEnv *env, jclass c, jlong h, jint k, jint nq, jintArray queryIids, jintArray iidOut,
                                      jdoubleArray distOut, jintArray countOut) {
    jint *q, *ii, *cc;
    jdouble *dd;
    (void)c;
    if (bad_len(env, queryIids, nq, 0, "queryIids") || bad_len(env, iidOut, (int64_t)nq * k, 0, "iidOut") ||
        bad_len(env, distOut, (int64_t)nq * k, 0, "distOut") || bad_len(env, countOut, nq, 0, "countOut"))
        return;
    q = (*env)->GetIntArrayElements(env, queryIids, NULL);
    ii = (*env)->GetIntArrayElements(env, iidOut, NULL);
    dd = (*env)->GetDoubleArrayElements(env, distOut, NULL);
    cc = (*env)->GetIntArrayElements(env, countOut, NULL);
    CHECK(mmidx_search_sdc(H(h), k, nq, (const int32_t *)q, (int32_t *)ii, dd, (int32_t *)cc));
done:
    (*env)->ReleaseIntArrayElements(env, queryIids, q, JNI_ABORT);
    (*env)->ReleaseIntArrayElements(env, iidOut, ii, 0);
    (*env)->ReleaseDoubleArrayElements(env, distOut, dd, 0);
    (*env)->ReleaseIntArrayElements(env, countOut, cc, 0);
}

JNIEXPORT void JNICALL JFN(listSizes)(JNIEnv *env, jclass c, jlong h, jintArray out) {
    dims_t d;
    jint *o;
    (void)c;
    if (get_dims(env, h, &d) || bad_len(env, out, d.C > 0 ? d.C : 1, 0, "out")) return;
    o = (*env)->GetIntArrayElements(env, out, NULL);
    CHECK(mmidx_list_sizes(H(h), (int32_t *)o));
done:
    (*env)->ReleaseIntArrayElements(env, out, o, 0);
}

/* native flat snapshot (mmidx_save / mmidx_load, ABI 8): the fast-restart path next to loadIndexInMemory's BDB cursor
 * (IVFPQ.java:680-728, PQ.java:436-483).  The id map stays in BDB. */
JNIEXPORT void JNICALL JFN(saveSnapshot)(JNIEnv *env, jclass c, jlong h, jstring path) {
    const char *p;
    (void)c;
    if (!path) {
        throw_status(env, MMIDX_ERR_INVALID_ARG);
        return;
    }
    p = (*env)->GetStringUTFChars(env, path, NULL);
    if (!p) return; /* OutOfMemoryError pending */
    CHECK(mmidx_save(H(h), p));
done:
    (*env)->ReleaseStringUTFChars(env, path, p);
}

JNIEXPORT void JNICALL JFN(loadSnapshot)(JNIEnv *env, jclass c, jlong h, jstring path) {
    const char *p;
    (void)c;
    if (!path) {
        throw_status(env, MMIDX_ERR_INVALID_ARG);
        return;
    }
    p = (*env)->GetStringUTFChars(env, path, NULL);
    if (!p) return;
    CHECK(mmidx_load(H(h), p));
done:
    (*env)->ReleaseStringUTFChars(env, path, p);
}

/* outputIndexingTimesInternal (ASS:718-729): the native side's accumulated statistics, as
 * {total_ms, coarse_ms, scan_ms, merge_ms, scan_codes, scan_launches, tie_fallbacks} */
JNIEXPORT void JNICALL JFN(stats)(JNIEnv *env, jclass c, jlong h, jdoubleArray out7) {
    mmidx_stats s;
    jdouble v[7];
    (void)c;
    if (bad_len(env, out7, 7, 0, "out")) return;
    CHECK(mmidx_get_stats(H(h), &s));
    v[0] = s.total_ms, v[1] = s.coarse_ms, v[2] = s.scan_ms, v[3] = s.merge_ms;
    v[4] = (double)s.scan_codes, v[5] = (double)s.scan_launches, v[6] = (double)s.tie_fallbacks;
    (*env)->SetDoubleArrayRegion(env, out7, 0, 7, v);
done:
    return;
}
JNIEXPORT void JNICALL JFN(setProfiling)(JNIEnv *env, jclass c, jlong h, jint on) {
    (void)c;
    CHECK(mmidx_set_profiling(H(h), on));
done:
    return;
}

/* ---- Linear ---------------------------------------------------------------------------------- */
#define HL(handle) ((mmidx_linear *)(intptr_t)(handle))

JNIEXPORT jlong JNICALL JFN(linearCreate)(JNIEnv *env, jclass c, jint D, jlong cap, jint device) {
    mmidx_linear *l = NULL;
    (void)c;
    CHECK(mmidx_linear_create(D, cap, device, &l));
done:
    return (jlong)(intptr_t)l;
}

JNIEXPORT void JNICALL JFN(linearDestroy)(JNIEnv *env, jclass c, jlong h) {
    (void)env;
    (void)c;
    mmidx_linear_destroy(HL(h));
}

JNIEXPORT void JNICALL JFN(linearAdd)(JNIEnv *env, jclass c, jlong h, jint n, jint D, jdoubleArray flat) {
    jdouble *a;
    (void)c;
    {   /* the vector length is the native object's, not the caller's word for it */
        int dn = 0;
        if (mmidx_linear_get_dim(HL(h), &dn) != MMIDX_OK || dn != D) {
            throw_msg(env, "java/lang/Exception", "The dimensionality of the vector is wrong!"); /* Linear.java:113 */
            return;
        }
    }
    if (n < 0 || bad_len(env, flat, (int64_t)n * D, 0, "vectors")) return;
    a = (*env)->GetDoubleArrayElements(env, flat, NULL);
    CHECK(mmidx_linear_add(HL(h), n, a));
done:
    (*env)->ReleaseDoubleArrayElements(env, flat, a, JNI_ABORT);
}

JNIEXPORT void JNICALL JFN(linearSearch)(JNIEnv *env, jclass c, jlong h, jint k, jint nq, jint D, jdoubleArray queries, jintArray iidOut,
                                         jdoubleArray distOut, jintArray countOut) {
    jdouble *q, *dd;
    jint *ii, *cc;
    (void)c;
    {
        int dn = 0;
        if (mmidx_linear_get_dim(HL(h), &dn) != MMIDX_OK || dn != D) {
            throw_msg(env, "java/lang/Exception", "The dimensionality of the vector is wrong!");
            return;
        }
    }
    if (nq < 0 || k < 1 || bad_len(env, queries, (int64_t)nq * D, 0, "queries") || bad_len(env, iidOut, (int64_t)nq * k, 0, "iidOut") ||
        bad_len(env, distOut, (int64_t)nq * k, 0, "distOut") || bad_len(env, countOut, nq, 0, "countOut"))
        return;
    q = (*env)->GetDoubleArrayElements(env, queries, NULL);
    ii = (*env)->GetIntArrayElements(env, iidOut, NULL);
    dd = (*env)->GetDoubleArrayElements(env, distOut, NULL);
    cc = (*env)->GetIntArrayElements(env, countOut, NULL);
    CHECK(mmidx_linear_search(HL(h), k, nq, q, (int32_t *)ii, dd, (int32_t *)cc));
done:
    (*env)->ReleaseDoubleArrayElements(env, queries, q, JNI_ABORT);
    (*env)->ReleaseIntArrayElements(env, iidOut, ii, 0);
    (*env)->ReleaseDoubleArrayElements(env, distOut, dd, 0);
    (*env)->ReleaseIntArrayElements(env, countOut, cc, 0);
}

/* ---- front end: PCA projection (PCA.java:188-208, :257-318) and VLAD aggregation (VladAggregator.java:56-70,
 *      VladAggregatorMultipleVocabularies.java:84-101) ------------------------------------------------------------- */
JNIEXPORT jlong JNICALL JFN(pcaCreate)(JNIEnv *env, jclass c, jint nc, jint ss, jboolean whitening, jdoubleArray means, jdoubleArray eig,
                                       jdoubleArray vt, jint device) {
    mmidx_pca *p = NULL;
    jdouble *m_, *e_, *v_;
    (void)c;
    if (bad_len(env, means, ss, 0, "means") || bad_len(env, eig, nc, whitening ? 0 : 1, "eigenvalues") ||
        bad_len(env, vt, (int64_t)nc * ss, 0, "components"))
        return 0;
    m_ = (*env)->GetDoubleArrayElements(env, means, NULL);
    e_ = eig ? (*env)->GetDoubleArrayElements(env, eig, NULL) : NULL;
    v_ = (*env)->GetDoubleArrayElements(env, vt, NULL);
    CHECK(mmidx_pca_create(nc, ss, whitening ? 1 : 0, m_, e_, v_, device, &p));
done:
    (*env)->ReleaseDoubleArrayElements(env, means, m_, JNI_ABORT);
    if (e_) (*env)->ReleaseDoubleArrayElements(env, eig, e_, JNI_ABORT);
    (*env)->ReleaseDoubleArrayElements(env, vt, v_, JNI_ABORT);
    return (jlong)(intptr_t)p;
}
JNIEXPORT void JNICALL JFN(pcaDestroy)(JNIEnv *env, jclass c, jlong p) {
    (void)env;
    (void)c;
    mmidx_pca_destroy((mmidx_pca *)(intptr_t)p);
}
/* sampleToEigenSpace for n samples: x[n][ss] -> y[n][nc] */
JNIEXPORT void JNICALL JFN(pcaProject)(JNIEnv *env, jclass c, jlong p, jint n, jint ss, jint nc, jdoubleArray x, jdoubleArray y) {
    jdouble *x_, *y_;
    (void)c;
    {   /* sampleSize / numComponents are the native object's: a wrapper that disagrees must not size the copies */
        int nn = 0, sn = 0;
        if (mmidx_pca_get_dims((const mmidx_pca *)(intptr_t)p, &nn, &sn) != MMIDX_OK || nn != nc || sn != ss) {
            throw_msg(env, "java/lang/IllegalArgumentException", "sampleSize / numComponents differ from the PCA object's");
            return;
        }
    }
    if (n < 0 || bad_len(env, x, (int64_t)n * ss, 0, "samples") || bad_len(env, y, (int64_t)n * nc, 0, "projected")) return;
    x_ = (*env)->GetDoubleArrayElements(env, x, NULL);
    y_ = (*env)->GetDoubleArrayElements(env, y, NULL);
    CHECK(mmidx_pca_project((mmidx_pca *)(intptr_t)p, n, x_, y_));
done:
    (*env)->ReleaseDoubleArrayElements(env, x, x_, JNI_ABORT);
    (*env)->ReleaseDoubleArrayElements(env, y, y_, 0);
}
JNIEXPORT jlong JNICALL JFN(vladCreate)(JNIEnv *env, jclass c, jintArray ncent, jint dl, jdoubleArray codebooks, jboolean normalizationsOn,
                                        jint device) {
    mmidx_vlad *v = NULL;
    jint nvocab, *nc_;
    jdouble *cb_;
    int64_t total = 0;
    int i;
    (void)c;
    if (bad_len(env, ncent, 1, 0, "centroid counts")) return 0;
    nvocab = (*env)->GetArrayLength(env, ncent);
    nc_ = (*env)->GetIntArrayElements(env, ncent, NULL);
    for (i = 0; i < nvocab; i++) total += nc_[i];
    if (bad_len(env, codebooks, total * dl, 0, "codebooks")) {
        (*env)->ReleaseIntArrayElements(env, ncent, nc_, JNI_ABORT);
        return 0;
    }
    cb_ = (*env)->GetDoubleArrayElements(env, codebooks, NULL);
    CHECK(mmidx_vlad_create(nvocab, (const int32_t *)nc_, dl, cb_, normalizationsOn ? 1 : 0, device, &v));
done:
    (*env)->ReleaseIntArrayElements(env, ncent, nc_, JNI_ABORT);
    (*env)->ReleaseDoubleArrayElements(env, codebooks, cb_, JNI_ABORT);
    return (jlong)(intptr_t)v;
}
JNIEXPORT void JNICALL JFN(vladDestroy)(JNIEnv *env, jclass c, jlong v) {
    (void)env;
    (void)c;
    mmidx_vlad_destroy((mmidx_vlad *)(intptr_t)v);
}
JNIEXPORT jint JNICALL JFN(vladVectorLength)(JNIEnv *env, jclass c, jlong v) {
    int len = 0;
    (void)c;
    CHECK(mmidx_vlad_vector_length((const mmidx_vlad *)(intptr_t)v, &len));
done:
    return len;
}
/* aggregate for nimg images: descOff[nimg + 1] delimits each image's descriptors in descs[total][dl]; out[nimg][vectorLength].
 * pca != 0: ImageVectorization.transformToVector (aggregate, then sampleToEigenSpace), out[nimg][nc]. */
JNIEXPORT void JNICALL JFN(vladAggregate)(JNIEnv *env, jclass c, jlong v, jlong pca, jint dl, jint outLen, jlongArray descOff,
                                          jdoubleArray descs, jdoubleArray out) {
    jint nimg;
    jlong *off_;
    jdouble *d_, *o_;
    (void)c;
    if (bad_len(env, descOff, 1, 0, "descOff")) return;
    nimg = (*env)->GetArrayLength(env, descOff) - 1;
    off_ = (*env)->GetLongArrayElements(env, descOff, NULL);
    {   /* descriptor length and output length from the native objects; the offsets must be non-decreasing from 0 */
        int dn = 0, vn = 0, pn = 0, ps = 0, bad = 0;
        jint i;
        if (mmidx_vlad_descriptor_length((const mmidx_vlad *)(intptr_t)v, &dn) != MMIDX_OK || mmidx_vlad_vector_length((const mmidx_vlad *)(intptr_t)v, &vn) != MMIDX_OK ||
            (pca && mmidx_pca_get_dims((const mmidx_pca *)(intptr_t)pca, &pn, &ps) != MMIDX_OK))
            bad = 1;
        if (!bad && (dn != dl || (pca ? (pn != outLen || ps != vn) : vn != outLen))) bad = 1;
        if (!bad && off_[0] != 0) bad = 2;
        for (i = 0; !bad && i < nimg; i++)
            if (off_[i + 1] < off_[i]) bad = 2;
        if (bad) {
            throw_msg(env, "java/lang/IllegalArgumentException",
                      bad == 2 ? "descOff must start at 0 and be non-decreasing" : "descriptor / output lengths differ from the native objects'");
            (*env)->ReleaseLongArrayElements(env, descOff, off_, JNI_ABORT);
            return;
        }
    }
    if (bad_len(env, descs, (int64_t)off_[nimg] * dl, 0, "descriptors") || bad_len(env, out, (int64_t)nimg * outLen, 0, "out")) {
        (*env)->ReleaseLongArrayElements(env, descOff, off_, JNI_ABORT);
        return;
    }
    d_ = (*env)->GetDoubleArrayElements(env, descs, NULL);
    o_ = (*env)->GetDoubleArrayElements(env, out, NULL);
    if (pca) CHECK(mmidx_vectorize((mmidx_vlad *)(intptr_t)v, (mmidx_pca *)(intptr_t)pca, nimg, (const int64_t *)off_, d_, o_));
    else CHECK(mmidx_vlad_aggregate((mmidx_vlad *)(intptr_t)v, nimg, (const int64_t *)off_, d_, o_));
done:
    (*env)->ReleaseLongArrayElements(env, descOff, off_, JNI_ABORT);
    (*env)->ReleaseDoubleArrayElements(env, descs, d_, JNI_ABORT);
    (*env)->ReleaseDoubleArrayElements(env, out, o_, 0);
}
