/* SYNTAX-CHECK STAND-IN, NOT the JDK's jni.h.
 *
 * The build image has no JDK, so multimedia-indexing_amd/jni/mmidx_jni.c can never be compiled for real here.  This
 * header declares just enough of the JNI C interface (types and the JNIEnv function table entries the shim uses, with
 * the signatures of the JNI specification) for `gcc -fsyntax-only` to type-check the shim in tests/test_jni_shim_cpu.py.
 * Nothing is ever linked or run against it; a real build uses the JDK header (CMakeLists.txt: find_package(JNI)). */
#ifndef MMIDX_TEST_JNI_STUB_H
#define MMIDX_TEST_JNI_STUB_H
#include <stdint.h>
typedef int32_t jint;
typedef int64_t jlong;
typedef int8_t jbyte;
typedef int16_t jshort;
typedef double jdouble;
typedef uint8_t jboolean;
typedef jint jsize;
struct _jobject;
typedef struct _jobject *jobject;
typedef jobject jclass;
typedef jobject jarray;
typedef jarray jintArray;
typedef jarray jlongArray;
typedef jarray jbyteArray;
typedef jarray jshortArray;
typedef jarray jdoubleArray;
typedef jobject jstring;
#define JNIEXPORT __attribute__((visibility("default")))
#define JNICALL
#define JNI_ABORT 2
struct JNINativeInterface_;
typedef const struct JNINativeInterface_ *JNIEnv;
struct JNINativeInterface_ {
    jclass (*FindClass)(JNIEnv *, const char *);
    jint (*ThrowNew)(JNIEnv *, jclass, const char *);
    jsize (*GetArrayLength)(JNIEnv *, jarray);
    jint *(*GetIntArrayElements)(JNIEnv *, jintArray, jboolean *);
    jlong *(*GetLongArrayElements)(JNIEnv *, jlongArray, jboolean *);
    jbyte *(*GetByteArrayElements)(JNIEnv *, jbyteArray, jboolean *);
    jshort *(*GetShortArrayElements)(JNIEnv *, jshortArray, jboolean *);
    jdouble *(*GetDoubleArrayElements)(JNIEnv *, jdoubleArray, jboolean *);
    void (*ReleaseIntArrayElements)(JNIEnv *, jintArray, jint *, jint);
    void (*ReleaseLongArrayElements)(JNIEnv *, jlongArray, jlong *, jint);
    void (*ReleaseByteArrayElements)(JNIEnv *, jbyteArray, jbyte *, jint);
    void (*ReleaseShortArrayElements)(JNIEnv *, jshortArray, jshort *, jint);
    void (*ReleaseDoubleArrayElements)(JNIEnv *, jdoubleArray, jdouble *, jint);
    void (*SetIntArrayRegion)(JNIEnv *, jintArray, jsize, jsize, const jint *);
    void (*SetDoubleArrayRegion)(JNIEnv *, jdoubleArray, jsize, jsize, const jdouble *);
    const char *(*GetStringUTFChars)(JNIEnv *, jstring, jboolean *);
    void (*ReleaseStringUTFChars)(JNIEnv *, jstring, const char *);
};
#endif
