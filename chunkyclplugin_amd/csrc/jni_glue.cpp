// jni_glue.cpp — JNI bindings of include/chunky_hip.h for java/dev/thatredox/chunkynative/hip/HipNative.java.
//
// Built only where a JDK exists:  g++ -shared -fPIC -I$JAVA_HOME/include -I$JAVA_HOME/include/linux \
//     jni_glue.cpp -L.. -lchunky_hip -o libchunky_hip_jni.so
// This image has no jni.h, so the translation unit is empty here (and is not part of the library).
#if __has_include(<jni.h>)
#include <jni.h>

#include <vector>

#include "../../include/chunky_hip.h"

namespace {
void throw_last(JNIEnv* env) {
    jclass cls = env->FindClass("java/lang/RuntimeException");
    if (cls) env->ThrowNew(cls, chunky_last_error());
}
#define CHECK(expr)                 \
    do {                            \
        if ((expr) != CHUNKY_OK) {  \
            throw_last(env);        \
            return;                 \
        }                           \
    } while (0)
struct Ints {  // pins a Java int[] for the duration of one call (the C side copies before returning)
    JNIEnv* env;
    jintArray arr;
    jint* p;
    jsize n;
    Ints(JNIEnv* e, jintArray a) : env(e), arr(a), p(a ? e->GetIntArrayElements(a, nullptr) : nullptr), n(a ? e->GetArrayLength(a) : 0) {}
    ~Ints() { if (p) env->ReleaseIntArrayElements(arr, p, JNI_ABORT); }
};
struct PostRender {
    JNIEnv* env;
    jobject supplier;
    jmethodID get;
};
int post_render_trampoline(void* user) {
    PostRender* pr = static_cast<PostRender*>(user);
    return pr->env->CallBooleanMethod(pr->supplier, pr->get) ? 1 : 0;
}
}  // namespace

#define J(name) Java_dev_thatredox_chunkynative_hip_HipNative_##name
extern "C" {
JNIEXPORT jint JNICALL J(deviceCount)(JNIEnv*, jclass) { return chunky_device_count(); }
JNIEXPORT jlong JNICALL J(init)(JNIEnv* env, jclass, jint device) {
    chunky_ctx* c = nullptr;
    if (chunky_init(device, &c) != CHUNKY_OK) throw_last(env);
    return (jlong)c;
}
JNIEXPORT void JNICALL J(shutdown)(JNIEnv* env, jclass, jlong ctx) { CHECK(chunky_shutdown((chunky_ctx*)ctx)); }
JNIEXPORT jlong JNICALL J(sceneCreate)(JNIEnv* env, jclass, jlong ctx) {
    chunky_scene* s = nullptr;
    if (chunky_scene_create((chunky_ctx*)ctx, &s) != CHUNKY_OK) throw_last(env);
    return (jlong)s;
}
JNIEXPORT void JNICALL J(sceneDestroy)(JNIEnv* env, jclass, jlong s) { CHECK(chunky_scene_destroy((chunky_scene*)s)); }
JNIEXPORT void JNICALL J(sceneLoadOctree)(JNIEnv* env, jclass, jlong s, jintArray tree, jint depth, jintArray mapping) {
    Ints t(env, tree), m(env, mapping);
    CHECK(chunky_scene_load_octree((chunky_scene*)s, (const int32_t*)t.p, t.n, depth, (const int32_t*)m.p, m.n));
}
JNIEXPORT void JNICALL J(sceneSetPalette)(JNIEnv* env, jclass, jlong s, jint kind, jintArray data) {
    Ints d(env, data);
    CHECK(chunky_scene_set_palette((chunky_scene*)s, kind, (const int32_t*)d.p, d.n));
}
JNIEXPORT void JNICALL J(sceneSetBvh)(JNIEnv* env, jclass, jlong s, jint which, jintArray nodes) {
    Ints d(env, nodes);
    CHECK(chunky_scene_set_bvh((chunky_scene*)s, which, (const int32_t*)d.p, d.n));
}
JNIEXPORT void JNICALL J(sceneSetAtlas)(JNIEnv* env, jclass, jlong s, jint w, jint h, jint layers) {
    CHECK(chunky_scene_set_atlas((chunky_scene*)s, nullptr, w, h, layers));
}
JNIEXPORT void JNICALL J(sceneWriteAtlasTile)(JNIEnv* env, jclass, jlong s, jint x, jint y, jint layer, jint w, jint h, jbyteArray rgba) {
    jbyte* p = env->GetByteArrayElements(rgba, nullptr);
    int rc = chunky_scene_write_atlas_tile((chunky_scene*)s, x, y, layer, w, h, (const uint8_t*)p);
    env->ReleaseByteArrayElements(rgba, p, JNI_ABORT);
    CHECK(rc);
}
JNIEXPORT void JNICALL J(sceneSetSky)(JNIEnv* env, jclass, jlong s, jbyteArray rgba, jint w, jint h, jfloat intensity) {
    jbyte* p = env->GetByteArrayElements(rgba, nullptr);
    int rc = chunky_scene_set_sky((chunky_scene*)s, (const uint8_t*)p, w, h, intensity);
    env->ReleaseByteArrayElements(rgba, p, JNI_ABORT);
    CHECK(rc);
}
JNIEXPORT void JNICALL J(sceneSetSun)(JNIEnv* env, jclass, jlong s, jintArray sun) {
    Ints d(env, sun);
    CHECK(chunky_scene_set_sun((chunky_scene*)s, (const int32_t*)d.p));
}
JNIEXPORT jlong JNICALL J(renderCreate)(JNIEnv* env, jclass, jlong ctx, jlong scene, jint w, jint h) {
    chunky_render* r = nullptr;
    if (chunky_render_create((chunky_ctx*)ctx, (chunky_scene*)scene, w, h, &r) != CHUNKY_OK) throw_last(env);
    return (jlong)r;
}
JNIEXPORT void JNICALL J(renderDestroy)(JNIEnv* env, jclass, jlong r) { CHECK(chunky_render_destroy((chunky_render*)r)); }
JNIEXPORT void JNICALL J(renderSetCamera)(JNIEnv* env, jclass, jlong r, jint type, jfloatArray settings) {
    jfloat* p = env->GetFloatArrayElements(settings, nullptr);
    int rc = chunky_render_set_camera((chunky_render*)r, type, p, env->GetArrayLength(settings));
    env->ReleaseFloatArrayElements(settings, p, JNI_ABORT);
    CHECK(rc);
}
JNIEXPORT void JNICALL J(renderPasses)(JNIEnv* env, jclass, jlong r, jintArray seeds, jint first) {
    Ints d(env, seeds);
    CHECK(chunky_render_passes((chunky_render*)r, (const int32_t*)d.p, d.n, first));
}
JNIEXPORT void JNICALL J(renderRead)(JNIEnv* env, jclass, jlong r, jfloatArray out) {
    jfloat* p = env->GetFloatArrayElements(out, nullptr);
    int rc = chunky_render_read((chunky_render*)r, p, env->GetArrayLength(out));
    env->ReleaseFloatArrayElements(out, p, 0);
    CHECK(rc);
}
JNIEXPORT void JNICALL J(renderPreview)(JNIEnv* env, jclass, jlong r, jintArray out) {
    jint* p = env->GetIntArrayElements(out, nullptr);
    int rc = chunky_render_preview((chunky_render*)r, (int32_t*)p);
    env->ReleaseIntArrayElements(out, p, 0);
    CHECK(rc);
}
JNIEXPORT void JNICALL J(filterFrame)(JNIEnv* env, jclass, jlong ctx, jint width, jint height, jdouble exposure,
                                      jdoubleArray input, jintArray out, jint type) {
    jdouble* in = env->GetDoubleArrayElements(input, nullptr);
    jint* p = env->GetIntArrayElements(out, nullptr);
    int rc = CHUNKY_E_INVALID;
    if ((jlong)env->GetArrayLength(input) >= 3LL * width * height && (jlong)env->GetArrayLength(out) >= (jlong)width * height)
        rc = chunky_filter_frame((chunky_ctx*)ctx, width, height, exposure, in, (int32_t*)p, type);
    env->ReleaseDoubleArrayElements(input, in, JNI_ABORT);
    env->ReleaseIntArrayElements(out, p, 0);
    CHECK(rc);
}
JNIEXPORT jint JNICALL J(renderRun)(JNIEnv* env, jclass, jlong r, jdoubleArray samples, jint sceneSpp, jint target,
                                    jint interval, jobject supplier) {
    PostRender pr{env, supplier, nullptr};
    if (supplier) pr.get = env->GetMethodID(env->GetObjectClass(supplier), "getAsBoolean", "()Z");
    jdouble* p = env->GetDoubleArrayElements(samples, nullptr);
    int32_t spp = sceneSpp;
    int rc = chunky_render_run((chunky_render*)r, p, &spp, target, interval, supplier ? post_render_trampoline : nullptr, &pr);
    env->ReleaseDoubleArrayElements(samples, p, 0);
    if (rc != CHUNKY_OK && rc != CHUNKY_E_ABORTED) throw_last(env);
    return spp;
}
}
#endif
