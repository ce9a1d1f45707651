// tests/jni_stub/fake_jvm.cpp — TEST-ONLY: a stand-in for the JVM side of csrc/jni_glue.cpp, so that the glue's native methods
// are not only parsed but RUN (tests/test_jni_glue_runs.py).  It implements the JNIEnv members tests/jni_stub/jni.h declares over
// fake Java objects, and then does what java/.../HipSceneLoader.java, HipPathTracingRenderer.java and HipPreviewRenderer.java do
// through HipNative: upload a packed scene, set the camera, run the pass loop with a RunListener, read a preview.
//
// The fake follows the JNI specification where a mistake in the glue would show:
//   * Get<T>ArrayElements hands out a COPY (as a copying collector does); Release<T>ArrayElements copies back for mode 0 and
//     JNI_COMMIT and discards for JNI_ABORT — data the glue releases with the wrong mode never reaches the "Java" array;
//   * ThrowNew leaves an exception pending, ExceptionCheck reports it; the driver looks at it after every native call, as the
//     interpreter would on return;
//   * GetMethodID fails (and leaves NoSuchMethodError pending) for a name / signature HipNative.RunListener does not declare;
//   * every Get has to be matched by a Release, every array access stays inside its array: violations are counted and reported.
// It is not a JVM: no threads, no GC, no local-reference tables.  Nothing here is shipped or linked into the product.
//
//   fake_jvm scene.raw out.f64 <target spp> <merge interval> [stop-after-polls N] [save-at SPP] [short-arrays]
#include <jni.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <type_traits>
#include <vector>

struct _jmethodID {
    std::string name, sig;
};

namespace {
struct ArrayInfo {
    std::vector<unsigned char> bytes;
    size_t elem = 1;
    int outstanding = 0;  // Get...Elements without Release
};
std::map<const void*, ArrayInfo> g_arrays;
std::map<const void*, void*> g_copies;  // element copy handed out -> its array
struct Pending {
    bool set = false;
    std::string cls, msg;
} g_exc;
int g_violations = 0;
std::vector<std::string> g_thrown;  // every exception the glue raised, in order

struct Listener : _jobject {
    int polls = 0, gates = 0, progress_calls = 0, regen = 0, last_progress = 0;
    std::vector<int> merges;
    int stop_after_polls = -1, save_at = -1;
    bool progress_monotonic = true;
    jdoubleArray samples = nullptr;
    double first_sample_at_first_merge = 0;  // what the "Java" array held when merged() ran: the glue must have copied by then
};
_jclass g_listener_class, g_other_class;
const void* g_listener = nullptr;
std::map<std::string, _jmethodID> g_methods;

template <typename T, typename A>
A new_array(size_t n, const T* init = nullptr) {
    auto* obj = new typename std::remove_pointer<A>::type();
    ArrayInfo& info = g_arrays[obj];
    info.elem = sizeof(T);
    info.bytes.assign(n * sizeof(T), 0);
    if (init && n) memcpy(info.bytes.data(), init, n * sizeof(T));
    return obj;
}
ArrayInfo* info_of(const void* a, const char* who) {
    auto it = g_arrays.find(a);
    if (it == g_arrays.end()) {
        fprintf(stderr, "fake_jvm: %s on something that is not an array\n", who);
        g_violations++;
        return nullptr;
    }
    return &it->second;
}
template <typename T>
T* get_elements(const void* a, const char* who) {
    ArrayInfo* info = info_of(a, who);
    if (!info) return nullptr;
    info->outstanding++;
    T* copy = (T*)malloc(info->bytes.size() ? info->bytes.size() : 1);
    memcpy(copy, info->bytes.data(), info->bytes.size());
    g_copies[copy] = const_cast<void*>(a);
    return copy;
}
void release_elements(const void* a, void* p, jint mode, const char* who) {
    ArrayInfo* info = info_of(a, who);
    auto it = g_copies.find(p);
    if (!info || it == g_copies.end() || it->second != a) {
        fprintf(stderr, "fake_jvm: %s of elements that were not handed out for this array\n", who);
        g_violations++;
        return;
    }
    if (mode != JNI_ABORT) memcpy(info->bytes.data(), p, info->bytes.size());
    if (mode != JNI_COMMIT) {
        info->outstanding--;
        g_copies.erase(it);
        free(p);
    }
}
}  // namespace

// ---- the JNIEnv members of tests/jni_stub/jni.h ------------------------------------------------------------------------------
jclass JNIEnv::FindClass(const char* name) {
    static std::map<std::string, _jclass> classes;
    return &classes[name];
}
jint JNIEnv::ThrowNew(jclass cls, const char* msg) {
    static const char* names[] = {"java/lang/RuntimeException", "java/lang/IllegalArgumentException"};
    std::string cname = "?";
    for (const char* n : names)
        if (FindClass(n) == cls) cname = n;
    if (g_exc.set) {
        fprintf(stderr, "fake_jvm: ThrowNew with an exception already pending\n");
        g_violations++;
    }
    g_exc = Pending{true, cname, msg ? msg : ""};
    g_thrown.push_back(cname + ": " + g_exc.msg);
    return 0;
}
jboolean JNIEnv::ExceptionCheck() { return g_exc.set; }
jclass JNIEnv::GetObjectClass(jobject obj) { return obj == g_listener ? &g_listener_class : &g_other_class; }
jmethodID JNIEnv::GetMethodID(jclass cls, const char* name, const char* sig) {
    // HipNative.RunListener (java/.../HipNative.java): the only class the glue looks methods up on
    static const std::map<std::string, std::string> declared = {{"postRender", "()Z"}, {"pollGate", "()Z"}, {"progress", "(I)V"},
                                                                 {"merged", "(I)V"},   {"saveEvent", "(I)I"}, {"regenerateCamera", "()V"}};
    auto it = declared.find(name);
    if (cls != &g_listener_class || it == declared.end() || it->second != sig) {
        g_exc = Pending{true, "java/lang/NoSuchMethodError", name};
        g_thrown.push_back(std::string("java/lang/NoSuchMethodError: ") + name);
        return nullptr;
    }
    _jmethodID& m = g_methods[name];
    m.name = name;
    m.sig = sig;
    return &m;
}
static Listener* listener_of(jobject obj, jmethodID m, const char* kind) {
    if (g_exc.set) {
        fprintf(stderr, "fake_jvm: Call%sMethod(%s) with an exception pending\n", kind, m ? m->name.c_str() : "?");
        g_violations++;
    }
    return static_cast<Listener*>(obj);
}
jboolean JNIEnv::CallBooleanMethod(jobject obj, jmethodID m, ...) {
    Listener* L = listener_of(obj, m, "Boolean");
    if (m->name == "postRender") {
        L->polls++;
        return L->stop_after_polls >= 0 && L->polls > L->stop_after_polls;
    }
    if (m->name == "pollGate") {
        L->gates++;
        return 1;
    }
    g_violations++;
    return 0;
}
jint JNIEnv::CallIntMethod(jobject obj, jmethodID m, ...) {
    Listener* L = listener_of(obj, m, "Int");
    va_list ap;
    va_start(ap, m);
    const jint spp = va_arg(ap, jint);
    va_end(ap);
    if (m->name != "saveEvent") g_violations++;
    return spp == L->save_at ? 1 : 0;
}
void JNIEnv::CallVoidMethod(jobject obj, jmethodID m, ...) {
    Listener* L = listener_of(obj, m, "Void");
    va_list ap;
    va_start(ap, m);
    if (m->name == "progress") {
        const jint spp = va_arg(ap, jint);
        if (spp <= L->last_progress) L->progress_monotonic = false;
        L->last_progress = spp;
        L->progress_calls++;
    } else if (m->name == "merged") {
        const jint spp = va_arg(ap, jint);
        if (L->merges.empty() && L->samples) {
            const ArrayInfo& a = g_arrays[L->samples];
            double acc = 0;
            for (size_t i = 0; i + 8 <= a.bytes.size(); i += 8) {
                double v;
                memcpy(&v, &a.bytes[i], 8);
                acc += v;
            }
            L->first_sample_at_first_merge = acc;
        }
        L->merges.push_back(spp);
    } else if (m->name == "regenerateCamera") {
        L->regen++;
    } else {
        g_violations++;
    }
    va_end(ap);
}
jstring JNIEnv::NewStringUTF(const char* utf) {
    static std::map<_jstring*, std::string> strings;
    auto* s = new _jstring();
    strings[s] = utf ? utf : "";
    printf("{\"string\": \"%s\"}\n", strings[s].c_str());
    return s;
}
jsize JNIEnv::GetArrayLength(jarray a) {
    ArrayInfo* info = info_of(a, "GetArrayLength");
    return info ? (jsize)(info->bytes.size() / info->elem) : 0;
}
jint* JNIEnv::GetIntArrayElements(jintArray a, jboolean* c) {
    if (c) *c = 1;
    return get_elements<jint>(a, "GetIntArrayElements");
}
jbyte* JNIEnv::GetByteArrayElements(jbyteArray a, jboolean* c) {
    if (c) *c = 1;
    return get_elements<jbyte>(a, "GetByteArrayElements");
}
jfloat* JNIEnv::GetFloatArrayElements(jfloatArray a, jboolean* c) {
    if (c) *c = 1;
    return get_elements<jfloat>(a, "GetFloatArrayElements");
}
jdouble* JNIEnv::GetDoubleArrayElements(jdoubleArray a, jboolean* c) {
    if (c) *c = 1;
    return get_elements<jdouble>(a, "GetDoubleArrayElements");
}
void JNIEnv::ReleaseIntArrayElements(jintArray a, jint* p, jint mode) { release_elements(a, p, mode, "ReleaseIntArrayElements"); }
void JNIEnv::ReleaseByteArrayElements(jbyteArray a, jbyte* p, jint mode) { release_elements(a, p, mode, "ReleaseByteArrayElements"); }
void JNIEnv::ReleaseFloatArrayElements(jfloatArray a, jfloat* p, jint mode) { release_elements(a, p, mode, "ReleaseFloatArrayElements"); }
void JNIEnv::ReleaseDoubleArrayElements(jdoubleArray a, jdouble* p, jint mode) { release_elements(a, p, mode, "ReleaseDoubleArrayElements"); }
void JNIEnv::GetDoubleArrayRegion(jdoubleArray a, jsize start, jsize len, jdouble* buf) {
    ArrayInfo* info = info_of(a, "GetDoubleArrayRegion");
    if (!info || start < 0 || len < 0 || (size_t)(start + len) * 8 > info->bytes.size()) {
        g_violations++;  // ArrayIndexOutOfBoundsException in a JVM
        return;
    }
    memcpy(buf, info->bytes.data() + (size_t)start * 8, (size_t)len * 8);
}
void JNIEnv::SetDoubleArrayRegion(jdoubleArray a, jsize start, jsize len, const jdouble* buf) {
    ArrayInfo* info = info_of(a, "SetDoubleArrayRegion");
    if (!info || start < 0 || len < 0 || (size_t)(start + len) * 8 > info->bytes.size()) {
        g_violations++;
        return;
    }
    memcpy(info->bytes.data() + (size_t)start * 8, buf, (size_t)len * 8);
}

// ---- the natives (csrc/jni_glue.cpp), as javah would declare them ----------------------------------------------------------------
#define J(name) Java_dev_thatredox_chunkynative_hip_HipNative_##name
extern "C" {
jint J(deviceCount)(JNIEnv*, jclass);
jstring J(deviceName)(JNIEnv*, jclass, jint);
jlong J(init)(JNIEnv*, jclass, jint);
jlong J(groupCreate)(JNIEnv*, jclass, jintArray);
jint J(groupSize)(JNIEnv*, jclass, jlong);
void J(shutdown)(JNIEnv*, jclass, jlong);
jlong J(sceneCreate)(JNIEnv*, jclass, jlong);
void J(sceneDestroy)(JNIEnv*, jclass, jlong);
void J(sceneLoadOctree)(JNIEnv*, jclass, jlong, jintArray, jint, jintArray);
void J(sceneSetPalette)(JNIEnv*, jclass, jlong, jint, jintArray);
void J(sceneSetBvh)(JNIEnv*, jclass, jlong, jint, jintArray);
void J(sceneSetAtlas)(JNIEnv*, jclass, jlong, jint, jint, jint);
void J(sceneWriteAtlasTile)(JNIEnv*, jclass, jlong, jint, jint, jint, jint, jint, jbyteArray);
void J(sceneSetSky)(JNIEnv*, jclass, jlong, jbyteArray, jint, jint, jfloat);
void J(sceneSetSun)(JNIEnv*, jclass, jlong, jintArray);
jlong J(renderCreate)(JNIEnv*, jclass, jlong, jlong, jint, jint);
void J(renderDestroy)(JNIEnv*, jclass, jlong);
void J(renderSetCamera)(JNIEnv*, jclass, jlong, jint, jfloatArray);
void J(renderPasses)(JNIEnv*, jclass, jlong, jintArray, jint);
void J(renderRead)(JNIEnv*, jclass, jlong, jfloatArray);
void J(renderPreview)(JNIEnv*, jclass, jlong, jint, jint, jintArray);
jint J(renderRun)(JNIEnv*, jclass, jlong, jint, jint, jdoubleArray, jint, jint, jint, jobject);
}

// ---- the scene dump of chunkyclplugin_amd.scenes.save_raw ----------------------------------------------------------------------
struct Raw {
    int dtype = 0;
    std::vector<int64_t> dims;
    std::vector<unsigned char> bytes;
    int64_t count() const {
        int64_t n = 1;
        for (int64_t d : dims) n *= d;
        return n;
    }
};
static bool read_scene(const char* path, std::map<std::string, Raw>* out) {
    FILE* f = fopen(path, "rb");
    if (!f) return false;
    char magic[8];
    if (fread(magic, 1, 8, f) != 8 || memcmp(magic, "CHKSCN01", 8) != 0) return false;
    for (;;) {
        char name[16];
        if (fread(name, 1, 16, f) != 16) break;
        int32_t head[2];
        int64_t dims[4];
        if (fread(head, 4, 2, f) != 2 || fread(dims, 8, 4, f) != 4) return false;
        Raw a;
        a.dtype = head[0];
        a.dims.assign(dims, dims + head[1]);
        static const int width[4] = {4, 1, 4, 8};
        a.bytes.resize((size_t)a.count() * width[a.dtype]);
        if (!a.bytes.empty() && fread(a.bytes.data(), 1, a.bytes.size(), f) != a.bytes.size()) return false;
        (*out)[std::string(name, strnlen(name, 16))] = std::move(a);
    }
    fclose(f);
    return true;
}

// what the interpreter does when a native returns: a pending exception propagates
static bool threw(const char* where) {
    if (!g_exc.set) return false;
    printf("{\"exception_at\": \"%s\", \"class\": \"%s\", \"message\": \"%s\"}\n", where, g_exc.cls.c_str(), g_exc.msg.c_str());
    g_exc.set = false;
    return true;
}
#define OR_DIE(where)          \
    if (threw(where)) return 3

int main(int argc, char** argv) {
    if (argc < 5) {
        fprintf(stderr, "usage: %s scene.raw out.f64 <target spp> <merge interval> [stop-after-polls N] [save-at SPP] [short-arrays]\n", argv[0]);
        return 1;
    }
    std::map<std::string, Raw> sc;
    if (!read_scene(argv[1], &sc)) {
        fprintf(stderr, "cannot read %s\n", argv[1]);
        return 1;
    }
    Listener L;
    g_listener = &L;
    bool short_arrays = false;
    for (int i = 5; i < argc; i++) {
        if (!strcmp(argv[i], "stop-after-polls") && i + 1 < argc) L.stop_after_polls = atoi(argv[++i]);
        if (!strcmp(argv[i], "save-at") && i + 1 < argc) L.save_at = atoi(argv[++i]);
        if (!strcmp(argv[i], "short-arrays")) short_arrays = true;
    }
    const int target = atoi(argv[3]), interval = atoi(argv[4]);
    JNIEnv env_obj, *env = &env_obj;
    jclass cls = nullptr;

    printf("{\"devices\": %d}\n", (int)J(deviceCount)(env, cls));
    const jlong ctx = J(init)(env, cls, 0);
    OR_DIE("init");  // no GPU: RuntimeException carrying chunky_last_error()
    J(deviceName)(env, cls, 0);
    OR_DIE("deviceName");

    // HipSceneLoader: the packed arrays as Java arrays
    auto ints = [&](const char* name) { return new_array<jint, jintArray>((size_t)sc[name].count(), (const jint*)sc[name].bytes.data()); };
    const int32_t* meta = (const int32_t*)sc["meta"].bytes.data();
    const int depth = meta[0], projector = meta[1], width = meta[2], height = meta[3];
    const jlong scene = J(sceneCreate)(env, cls, ctx);
    OR_DIE("sceneCreate");
    J(sceneLoadOctree)(env, cls, scene, ints("octree"), depth, new_array<jint, jintArray>(0));  // already remapped: an empty blockMapping maps nothing
    OR_DIE("sceneLoadOctree");
    const char* palettes[5] = {"block_palette", "material_palette", "aabb_models", "quad_models", "bvh_trigs"};
    for (int k = 0; k < 5; k++) {
        J(sceneSetPalette)(env, cls, scene, k, ints(palettes[k]));
        OR_DIE("sceneSetPalette");
    }
    J(sceneSetBvh)(env, cls, scene, 0, ints("world_bvh"));
    OR_DIE("sceneSetBvh world");
    J(sceneSetBvh)(env, cls, scene, 1, ints("actor_bvh"));
    OR_DIE("sceneSetBvh actor");
    const Raw& atlas = sc["atlas"];  // [layers][H][W][4]: ClTextureLoader writes tile by tile; here one tile per layer
    const int aw = (int)atlas.dims[2], ah = (int)atlas.dims[1], layers = (int)atlas.dims[0];
    J(sceneSetAtlas)(env, cls, scene, aw, ah, layers);
    OR_DIE("sceneSetAtlas");
    for (int l = 0; l < layers; l++) {
        J(sceneWriteAtlasTile)(env, cls, scene, 0, 0, l, aw, ah, new_array<jbyte, jbyteArray>((size_t)4 * aw * ah, (const jbyte*)atlas.bytes.data() + (size_t)l * 4 * aw * ah));
        OR_DIE("sceneWriteAtlasTile");
    }
    const Raw& sky = sc["sky"];
    J(sceneSetSky)(env, cls, scene, new_array<jbyte, jbyteArray>(sky.bytes.size(), (const jbyte*)sky.bytes.data()), (jint)sky.dims[1], (jint)sky.dims[0],
                   *(const float*)sc["sky_intensity"].bytes.data());
    OR_DIE("sceneSetSky");
    J(sceneSetSun)(env, cls, scene, ints("sun"));
    OR_DIE("sceneSetSun");

    const jlong render = J(renderCreate)(env, cls, ctx, scene, width, height);
    OR_DIE("renderCreate");
    J(renderSetCamera)(env, cls, render, projector, new_array<jfloat, jfloatArray>((size_t)sc["camera"].count(), (const jfloat*)sc["camera"].bytes.data()));
    OR_DIE("renderSetCamera");

    if (short_arrays) {  // the JVM heap must not be overrun: each of these has to raise IllegalArgumentException BEFORE any C call
        J(sceneSetSun)(env, cls, scene, new_array<jint, jintArray>(5));
        threw("short sun");
        J(sceneSetSky)(env, cls, scene, new_array<jbyte, jbyteArray>(10), 4, 4, 1.0f);
        threw("short sky");
        J(renderPreview)(env, cls, render, width, height, new_array<jint, jintArray>((size_t)width * height - 1));
        threw("short preview");
        J(renderRun)(env, cls, render, width, height, new_array<jdouble, jdoubleArray>((size_t)3 * width * height - 1), 0, 1, 1, nullptr);
        threw("short sample buffer");
        J(renderRead)(env, cls, render, new_array<jfloat, jfloatArray>(7));  // right type, wrong length: the C side refuses -> RuntimeException
        threw("wrong-length read");
        J(sceneSetPalette)(env, cls, 0, 0, new_array<jint, jintArray>(2));   // a NULL handle -> RuntimeException, not a crash
        threw("null scene");
    }

    // HipPreviewRenderer
    jintArray argb = new_array<jint, jintArray>((size_t)width * height);
    J(renderPreview)(env, cls, render, width, height, argb);
    OR_DIE("renderPreview");
    unsigned long long preview_sum = 0;
    for (size_t i = 0; i < g_arrays[argb].bytes.size(); i += 4) {
        uint32_t v;
        memcpy(&v, &g_arrays[argb].bytes[i], 4);
        preview_sum += v;
    }

    // HipPathTracingRenderer.render: scene.getSampleBuffer(), the listener, the loop
    jdoubleArray samples = new_array<jdouble, jdoubleArray>((size_t)3 * width * height);
    L.samples = samples;
    const jint spp = J(renderRun)(env, cls, render, width, height, samples, 0, target, interval, &L);
    OR_DIE("renderRun");
    FILE* f = fopen(argv[2], "wb");
    if (!f || fwrite(g_arrays[samples].bytes.data(), 1, g_arrays[samples].bytes.size(), f) != g_arrays[samples].bytes.size()) return 1;
    fclose(f);

    // one plain launch + read-back through the float[] path (mode 0 on success: the data has to arrive in the "Java" array)
    jintArray seeds = new_array<jint, jintArray>(1);
    const jint seed0 = -1155484576;  // new java.util.Random(0).nextInt()
    memcpy(g_arrays[seeds].bytes.data(), &seed0, 4);
    J(renderPasses)(env, cls, render, seeds, 0);
    OR_DIE("renderPasses");
    jfloatArray fb = new_array<jfloat, jfloatArray>((size_t)3 * width * height);
    J(renderRead)(env, cls, render, fb);
    OR_DIE("renderRead");
    double fb_sum = 0;
    for (size_t i = 0; i < g_arrays[fb].bytes.size(); i += 4) {
        float v;
        memcpy(&v, &g_arrays[fb].bytes[i], 4);
        fb_sum += v;
    }

    J(renderDestroy)(env, cls, render);
    OR_DIE("renderDestroy");
    J(sceneDestroy)(env, cls, scene);
    OR_DIE("sceneDestroy");
    J(shutdown)(env, cls, ctx);
    OR_DIE("shutdown");

    int outstanding = 0;
    for (auto& kv : g_arrays) outstanding += kv.second.outstanding;
    printf("{\"spp\": %d, \"polls\": %d, \"gates\": %d, \"progress_calls\": %d, \"progress_monotonic\": %s, \"regenerate_calls\": %d, \"merges\": [", (int)spp,
           L.polls, L.gates, L.progress_calls, L.progress_monotonic ? "true" : "false", L.regen);
    for (size_t i = 0; i < L.merges.size(); i++) printf("%s%d", i ? ", " : "", L.merges[i]);
    printf("], \"array_nonzero_at_first_merge\": %s, \"preview_sum\": %llu, \"read_sum\": %.9g, \"unreleased_arrays\": %d, \"violations\": %d, \"thrown\": %zu}\n",
           L.first_sample_at_first_merge != 0 ? "true" : "false", preview_sum, fb_sum, outstanding, g_violations, g_thrown.size());
    return g_violations || outstanding ? 4 : 0;
}
