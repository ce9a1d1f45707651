/* TEST-ONLY minimal jni.h: just enough of the JNI C++ surface for `g++ -fsyntax-only` of
 * chunkyclplugin_amd/csrc/jni_glue.cpp (tests/test_jni_surface.py).  It declares types and member signatures as the
 * JNI specification gives them and implements nothing; it is never shipped, never linked, and no product build sees it
 * (the product's include path has no jni.h in this image, so jni_glue.cpp is an empty translation unit there). */
#ifndef TEST_STUB_JNI_H
#define TEST_STUB_JNI_H
#include <stdint.h>
#define JNIEXPORT __attribute__((visibility("default")))
#define JNICALL
#define JNI_COMMIT 1
#define JNI_ABORT 2
typedef int32_t jint;
typedef int64_t jlong;
typedef int8_t jbyte;
typedef uint8_t jboolean;
typedef float jfloat;
typedef double jdouble;
typedef jint jsize;
class _jobject {};
class _jclass : public _jobject {};
class _jstring : public _jobject {};
class _jarray : public _jobject {};
class _jintArray : public _jarray {};
class _jbyteArray : public _jarray {};
class _jfloatArray : public _jarray {};
class _jdoubleArray : public _jarray {};
typedef _jobject* jobject;
typedef _jclass* jclass;
typedef _jstring* jstring;
typedef _jarray* jarray;
typedef _jintArray* jintArray;
typedef _jbyteArray* jbyteArray;
typedef _jfloatArray* jfloatArray;
typedef _jdoubleArray* jdoubleArray;
struct _jmethodID;
typedef _jmethodID* jmethodID;
struct JNIEnv {
    jclass FindClass(const char* name);
    jint ThrowNew(jclass cls, const char* msg);
    jboolean ExceptionCheck();
    jclass GetObjectClass(jobject obj);
    jmethodID GetMethodID(jclass cls, const char* name, const char* sig);
    jboolean CallBooleanMethod(jobject obj, jmethodID m, ...);
    jint CallIntMethod(jobject obj, jmethodID m, ...);
    void CallVoidMethod(jobject obj, jmethodID m, ...);
    jstring NewStringUTF(const char* utf);
    jsize GetArrayLength(jarray a);
    jint* GetIntArrayElements(jintArray a, jboolean* is_copy);
    jbyte* GetByteArrayElements(jbyteArray a, jboolean* is_copy);
    jfloat* GetFloatArrayElements(jfloatArray a, jboolean* is_copy);
    jdouble* GetDoubleArrayElements(jdoubleArray a, jboolean* is_copy);
    void ReleaseIntArrayElements(jintArray a, jint* p, jint mode);
    void ReleaseByteArrayElements(jbyteArray a, jbyte* p, jint mode);
    void ReleaseFloatArrayElements(jfloatArray a, jfloat* p, jint mode);
    void ReleaseDoubleArrayElements(jdoubleArray a, jdouble* p, jint mode);
    void GetDoubleArrayRegion(jdoubleArray a, jsize start, jsize len, jdouble* buf);
    void SetDoubleArrayRegion(jdoubleArray a, jsize start, jsize len, const jdouble* buf);
};
#endif
