// jni_glue.cpp — JNI bindings of include/chunky_hip.h for java/dev/thatredox/chunkynative/hip/HipNative.java.
//
// Built only where a JDK exists:  g++ -shared -fPIC -I$JAVA_HOME/include -I$JAVA_HOME/include/linux
//     jni_glue.cpp -L.. -lchunky_hip -o libchunky_hip_jni.so
// This image has no JDK: the translation unit is empty in the product build.  tests/test_jni_surface.py checks it
// anyway — every `native` method of HipNative.java has an export here with the same arity, and the file passes
// `g++ -fsyntax-only` against a minimal test-only jni.h (tests/jni_stub/, never shipped, never linked).
//
// Conventions: a negative chunky status becomes RuntimeException(chunky_last_error()) (the OpenCL build gets JOCL's
// CLException in the same places, RendererInstance.java:36); an array that is too short for what the C side will read
// or write becomes IllegalArgumentException BEFORE the C call — the C ABI trusts sizes, the JVM heap must not.
#if __has_include(<jni.h>)
#include <jni.h>

#include <cstdio>
#include <vector>

#include "../../include/chunky_hip.h"

namespace {
void throw_last(JNIEnv* env) {
    jclass cls = env->FindClass("java/lang/RuntimeException");
    if (cls) env->ThrowNew(cls, chunky_last_error());
}
bool bad_length(JNIEnv* env, jarray arr, long long need, const char* what) {
    const long long have = arr ? (long long)env->GetArrayLength(arr) : -1;
    if (arr && need >= 0 && have >= need) return false;
    char msg[160];
    snprintf(msg, sizeof msg, "%s: array of %lld elements, %lld needed", what, have, need);
    jclass cls = env->FindClass("java/lang/IllegalArgumentException");
    if (cls) env->ThrowNew(cls, msg);
    return true;
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

// The listener of HipNative.renderRun (HipNative.RunListener): the six hooks of chunky_run_callbacks.  The loop merges
// into a NATIVE buffer; before `merged` reaches Java the buffer is copied into the Java double[] (SetDoubleArrayRegion),
// so scene.postProcessFrame sees the merged samples on copying and on pinning JVMs alike.
struct Run {
    JNIEnv* env;
    jobject listener;
    jmethodID post_render, progress, merged, save_event, regenerate_camera, poll_gate;
    jdoubleArray samples;
    std::vector<double>* buffer;
    bool failed;  // a Java exception is pending: stop the loop
};
bool pending(Run* run) {
    if (run->env->ExceptionCheck()) run->failed = true;
    return run->failed;
}
int cb_post_render(void* user) {
    Run* run = static_cast<Run*>(user);
    if (run->failed) return 1;
    const bool stop = run->env->CallBooleanMethod(run->listener, run->post_render);
    return (pending(run) || stop) ? 1 : 0;
}
void cb_progress(void* user, int32_t scene_spp) {
    Run* run = static_cast<Run*>(user);
    if (run->failed) return;
    run->env->CallVoidMethod(run->listener, run->progress, (jint)scene_spp);
    pending(run);
}
void cb_merged(void* user, int32_t sample_spp) {
    Run* run = static_cast<Run*>(user);
    if (run->failed) return;
    run->env->SetDoubleArrayRegion(run->samples, 0, (jsize)run->buffer->size(), run->buffer->data());
    run->env->CallVoidMethod(run->listener, run->merged, (jint)sample_spp);
    pending(run);
}
int cb_save_event(void* user, int32_t spp) {
    Run* run = static_cast<Run*>(user);
    if (run->failed) return 0;
    const jint due = run->env->CallIntMethod(run->listener, run->save_event, (jint)spp);
    return pending(run) ? 0 : (int)due;
}
int cb_poll_gate(void* user) {
    Run* run = static_cast<Run*>(user);
    if (run->failed) return 1;  // let the poll through: cb_post_render then stops the loop
    const bool open = run->env->CallBooleanMethod(run->listener, run->poll_gate);
    return (pending(run) || open) ? 1 : 0;
}
void cb_regenerate_camera(void* user) {
    Run* run = static_cast<Run*>(user);
    if (run->failed) return;
    run->env->CallVoidMethod(run->listener, run->regenerate_camera);
    pending(run);
}
}  // namespace

#define J(name) Java_dev_thatredox_chunkynative_hip_HipNative_##name
extern "C" {
JNIEXPORT jint JNICALL J(deviceCount)(JNIEnv*, jclass) { return chunky_device_count(); }
JNIEXPORT jstring JNICALL J(deviceName)(JNIEnv* env, jclass, jint device) {
    char name[256];
    if (chunky_device_name(device, name, sizeof name) != CHUNKY_OK) {
        throw_last(env);
        return nullptr;
    }
    return env->NewStringUTF(name);
}
JNIEXPORT jlong JNICALL J(init)(JNIEnv* env, jclass, jint device) {
    chunky_ctx* c = nullptr;
    if (chunky_init(device, &c) != CHUNKY_OK) throw_last(env);
    return (jlong)c;
}
JNIEXPORT jlong JNICALL J(groupCreate)(JNIEnv* env, jclass, jintArray devices) {
    if (bad_length(env, devices, 1, "groupCreate")) return 0;
    Ints d(env, devices);
    chunky_ctx* c = nullptr;
    if (chunky_group_create((const int*)d.p, d.n, &c) != CHUNKY_OK) throw_last(env);
    return (jlong)c;
}
JNIEXPORT jint JNICALL J(groupSize)(JNIEnv* env, jclass, jlong ctx) {
    const int n = chunky_group_size((chunky_ctx*)ctx);
    if (n < 0) throw_last(env);
    return n;
}
JNIEXPORT void JNICALL J(groupPeerStatus)(JNIEnv* env, jclass, jlong ctx, jintArray out) {
    const int n = chunky_group_size((chunky_ctx*)ctx);
    if (n < 0) {
        throw_last(env);
        return;
    }
    if (bad_length(env, out, n, "groupPeerStatus")) return;
    jint* p = env->GetIntArrayElements(out, nullptr);
    int rc = chunky_group_peer_status((chunky_ctx*)ctx, (int*)p, n);
    env->ReleaseIntArrayElements(out, p, rc == CHUNKY_OK ? 0 : JNI_ABORT);
    CHECK(rc);
}
JNIEXPORT jint JNICALL J(groupTransport)(JNIEnv* env, jclass, jlong ctx) {
    int t = 0;
    if (chunky_group_transport((chunky_ctx*)ctx, &t, nullptr, 0) != CHUNKY_OK) throw_last(env);
    return t;
}
JNIEXPORT jstring JNICALL J(groupTransportDetail)(JNIEnv* env, jclass, jlong ctx) {
    int t = 0;
    char detail[512];
    if (chunky_group_transport((chunky_ctx*)ctx, &t, detail, sizeof detail) != CHUNKY_OK) {
        throw_last(env);
        return nullptr;
    }
    return env->NewStringUTF(detail);
}
JNIEXPORT void JNICALL J(groupSetTransport)(JNIEnv* env, jclass, jlong ctx, jint transport) {
    CHECK(chunky_group_set_transport((chunky_ctx*)ctx, transport));
}
JNIEXPORT void JNICALL J(shutdown)(JNIEnv* env, jclass, jlong ctx) { CHECK(chunky_shutdown((chunky_ctx*)ctx)); }
JNIEXPORT jlong JNICALL J(sceneCreate)(JNIEnv* env, jclass, jlong ctx) {
    chunky_scene* s = nullptr;
    if (chunky_scene_create((chunky_ctx*)ctx, &s) != CHUNKY_OK) throw_last(env);
    return (jlong)s;
}
JNIEXPORT void JNICALL J(sceneDestroy)(JNIEnv* env, jclass, jlong s) { CHECK(chunky_scene_destroy((chunky_scene*)s)); }
JNIEXPORT void JNICALL J(sceneLoadOctree)(JNIEnv* env, jclass, jlong s, jintArray tree, jint depth, jintArray mapping) {
    if (bad_length(env, tree, 1, "sceneLoadOctree tree") || bad_length(env, mapping, 0, "sceneLoadOctree mapping")) return;
    Ints t(env, tree), m(env, mapping);
    CHECK(chunky_scene_load_octree((chunky_scene*)s, (const int32_t*)t.p, t.n, depth, (const int32_t*)m.p, m.n));
}
JNIEXPORT void JNICALL J(sceneSetPalette)(JNIEnv* env, jclass, jlong s, jint kind, jintArray data) {
    if (bad_length(env, data, 0, "sceneSetPalette")) return;
    Ints d(env, data);
    CHECK(chunky_scene_set_palette((chunky_scene*)s, kind, (const int32_t*)d.p, d.n));
}
JNIEXPORT void JNICALL J(sceneSetBvh)(JNIEnv* env, jclass, jlong s, jint which, jintArray nodes) {
    if (bad_length(env, nodes, 7, "sceneSetBvh")) return;
    Ints d(env, nodes);
    CHECK(chunky_scene_set_bvh((chunky_scene*)s, which, (const int32_t*)d.p, d.n));
}
JNIEXPORT void JNICALL J(sceneSetAtlas)(JNIEnv* env, jclass, jlong s, jint w, jint h, jint layers) {
    CHECK(chunky_scene_set_atlas((chunky_scene*)s, nullptr, w, h, layers));
}
JNIEXPORT void JNICALL J(sceneWriteAtlasTile)(JNIEnv* env, jclass, jlong s, jint x, jint y, jint layer, jint w, jint h, jbyteArray rgba) {
    if (w <= 0 || h <= 0 || bad_length(env, rgba, 4LL * w * h, "sceneWriteAtlasTile")) {
        if (!env->ExceptionCheck()) bad_length(env, nullptr, 1, "sceneWriteAtlasTile size");
        return;
    }
    jbyte* p = env->GetByteArrayElements(rgba, nullptr);
    int rc = chunky_scene_write_atlas_tile((chunky_scene*)s, x, y, layer, w, h, (const uint8_t*)p);
    env->ReleaseByteArrayElements(rgba, p, JNI_ABORT);
    CHECK(rc);
}
JNIEXPORT void JNICALL J(sceneSetSky)(JNIEnv* env, jclass, jlong s, jbyteArray rgba, jint w, jint h, jfloat intensity) {
    if (w <= 0 || h <= 0 || bad_length(env, rgba, 4LL * w * h, "sceneSetSky")) {
        if (!env->ExceptionCheck()) bad_length(env, nullptr, 1, "sceneSetSky size");
        return;
    }
    jbyte* p = env->GetByteArrayElements(rgba, nullptr);
    int rc = chunky_scene_set_sky((chunky_scene*)s, (const uint8_t*)p, w, h, intensity);
    env->ReleaseByteArrayElements(rgba, p, JNI_ABORT);
    CHECK(rc);
}
JNIEXPORT void JNICALL J(sceneSetSun)(JNIEnv* env, jclass, jlong s, jintArray sun) {
    if (bad_length(env, sun, 6, "sceneSetSun")) return;
    Ints d(env, sun);
    CHECK(chunky_scene_set_sun((chunky_scene*)s, (const int32_t*)d.p));
}
JNIEXPORT jlong JNICALL J(renderCreate)(JNIEnv* env, jclass, jlong ctx, jlong scene, jint w, jint h) {
    chunky_render* r = nullptr;
    if (chunky_render_create((chunky_ctx*)ctx, (chunky_scene*)scene, w, h, &r) != CHUNKY_OK) throw_last(env);
    return (jlong)r;
}
JNIEXPORT void JNICALL J(renderDestroy)(JNIEnv* env, jclass, jlong r) { CHECK(chunky_render_destroy((chunky_render*)r)); }
JNIEXPORT void JNICALL J(renderSetOption)(JNIEnv* env, jclass, jlong r, jint option, jint value) {
    CHECK(chunky_render_set_option((chunky_render*)r, option, value));
}
JNIEXPORT void JNICALL J(renderSetCamera)(JNIEnv* env, jclass, jlong r, jint type, jfloatArray settings) {
    if (bad_length(env, settings, 15, "renderSetCamera")) return;  // the C side checks the exact count for the projector
    jfloat* p = env->GetFloatArrayElements(settings, nullptr);
    int rc = chunky_render_set_camera((chunky_render*)r, type, p, env->GetArrayLength(settings));
    env->ReleaseFloatArrayElements(settings, p, JNI_ABORT);
    CHECK(rc);
}
JNIEXPORT void JNICALL J(renderPasses)(JNIEnv* env, jclass, jlong r, jintArray seeds, jint first) {
    if (bad_length(env, seeds, 0, "renderPasses")) return;
    Ints d(env, seeds);
    CHECK(chunky_render_passes((chunky_render*)r, (const int32_t*)d.p, d.n, first));
}
JNIEXPORT void JNICALL J(renderRead)(JNIEnv* env, jclass, jlong r, jfloatArray out) {
    if (bad_length(env, out, 0, "renderRead")) return;  // chunky_render_read rejects any length but 3*width*height
    jfloat* p = env->GetFloatArrayElements(out, nullptr);
    int rc = chunky_render_read((chunky_render*)r, p, env->GetArrayLength(out));
    env->ReleaseFloatArrayElements(out, p, rc == CHUNKY_OK ? 0 : JNI_ABORT);
    CHECK(rc);
}
JNIEXPORT void JNICALL J(renderPreview)(JNIEnv* env, jclass, jlong r, jint width, jint height, jintArray out) {
    if (width <= 0 || height <= 0 || bad_length(env, out, (long long)width * height, "renderPreview")) {
        if (!env->ExceptionCheck()) bad_length(env, nullptr, 1, "renderPreview size");
        return;
    }
    // width/height are the render target's own (HipPreviewRenderer passes scene.width/height it created the target with)
    jint* p = env->GetIntArrayElements(out, nullptr);
    int rc = chunky_render_preview((chunky_render*)r, (int32_t*)p);
    env->ReleaseIntArrayElements(out, p, rc == CHUNKY_OK ? 0 : JNI_ABORT);
    CHECK(rc);
}
JNIEXPORT void JNICALL J(filterFrame)(JNIEnv* env, jclass, jlong ctx, jint width, jint height, jdouble exposure,
                                      jdoubleArray input, jintArray out, jint type) {
    if (width < 0 || height < 0 || bad_length(env, input, 3LL * width * height, "filterFrame input") ||
        bad_length(env, out, (long long)width * height, "filterFrame output")) {
        if (!env->ExceptionCheck()) bad_length(env, nullptr, 1, "filterFrame size");
        return;
    }
    jdouble* in = env->GetDoubleArrayElements(input, nullptr);
    jint* p = env->GetIntArrayElements(out, nullptr);
    int rc = chunky_filter_frame((chunky_ctx*)ctx, width, height, exposure, in, (int32_t*)p, type);
    env->ReleaseDoubleArrayElements(input, in, JNI_ABORT);
    env->ReleaseIntArrayElements(out, p, rc == CHUNKY_OK ? 0 : JNI_ABORT);
    CHECK(rc);
}
// chunky_render_run_ex.  `samples` is scene.getSampleBuffer(): read once at the start (GetDoubleArrayRegion), written at
// every merge; never pinned across the run.  Returns the new scene.spp.
JNIEXPORT jint JNICALL J(renderRun)(JNIEnv* env, jclass, jlong r, jint width, jint height, jdoubleArray samples, jint sceneSpp,
                                    jint target, jint interval, jobject listener) {
    if (width <= 0 || height <= 0 || bad_length(env, samples, 3LL * width * height, "renderRun sample buffer")) {
        if (!env->ExceptionCheck()) bad_length(env, nullptr, 1, "renderRun size");
        return sceneSpp;
    }
    std::vector<double> buffer((size_t)3 * width * height);
    env->GetDoubleArrayRegion(samples, 0, (jsize)buffer.size(), buffer.data());
    Run run{env, listener, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, samples, &buffer, false};
    chunky_run_callbacks cb{sizeof(chunky_run_callbacks), nullptr, nullptr, cb_merged, nullptr, nullptr, &run, nullptr};
    if (listener) {
        jclass cls = env->GetObjectClass(listener);
        run.post_render = env->GetMethodID(cls, "postRender", "()Z");
        run.progress = env->GetMethodID(cls, "progress", "(I)V");
        run.merged = env->GetMethodID(cls, "merged", "(I)V");
        run.save_event = env->GetMethodID(cls, "saveEvent", "(I)I");
        run.regenerate_camera = env->GetMethodID(cls, "regenerateCamera", "()V");
        run.poll_gate = env->GetMethodID(cls, "pollGate", "()Z");
        if (!run.post_render || !run.progress || !run.merged || !run.save_event || !run.regenerate_camera || !run.poll_gate) return sceneSpp;  // NoSuchMethodError pending
        cb.post_render = cb_post_render;
        cb.progress = cb_progress;
        cb.save_event = cb_save_event;
        cb.regenerate_camera = cb_regenerate_camera;
        cb.poll_gate = cb_poll_gate;
    } else {
        run.failed = true;  // no listener: nothing to call back; merges still land in the Java array at the end
    }
    int32_t spp = sceneSpp;
    int rc = chunky_render_run_ex((chunky_render*)r, buffer.data(), &spp, target, interval, &cb);
    if (!listener && !env->ExceptionCheck()) env->SetDoubleArrayRegion(samples, 0, (jsize)buffer.size(), buffer.data());
    if (rc != CHUNKY_OK && rc != CHUNKY_E_ABORTED && !env->ExceptionCheck()) throw_last(env);
    return spp;
}
}
#endif
