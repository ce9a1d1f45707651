package dev.thatredox.chunkynative.hip;

/**
 * JNI declarations for libchunky_hip.so (include/chunky_hip.h).  One native method per C entry
 * point the plugin needs; handles are opaque longs.  Every method throws RuntimeException carrying
 * chunky_last_error() when the C call returns a negative status (the OpenCL build throws JOCL's
 * CLException in the same places, RendererInstance.java:36).
 *
 * Written against the Chunky 2.5.0 plugin API without a JDK at hand: NOT compiled in this
 * repository (see INTEGRATION.md).
 */
public final class HipNative {
    static {
        System.loadLibrary("chunky_hip_jni"); // csrc/jni_glue.cpp, linked against libchunky_hip.so
    }

    private HipNative() {}

    // device — replaces RendererInstance.java:31-110
    public static native int deviceCount();
    public static native String deviceName(int device);
    public static native long init(int device);
    public static native void shutdown(long ctx);

    // scene — replaces ClIntBuffer / ClTextureLoader / ClSky uploads (ClSceneLoader.java:52-150)
    public static native long sceneCreate(long ctx);
    public static native void sceneDestroy(long scene);
    public static native void sceneLoadOctree(long scene, int[] treeData, int depth, int[] blockMapping);
    public static native void sceneSetPalette(long scene, int kind, int[] data);
    public static native void sceneSetBvh(long scene, int which, int[] nodes);
    public static native void sceneSetAtlas(long scene, int width, int height, int layers);
    public static native void sceneWriteAtlasTile(long scene, int x, int y, int layer, int w, int h, byte[] rgba);
    public static native void sceneSetSky(long scene, byte[] rgba, int width, int height, float intensity);
    public static native void sceneSetSun(long scene, int[] sun6);

    // render — replaces the buffers + launch loop of OpenClPathTracingRenderer.java:67-184
    public static native long renderCreate(long ctx, long scene, int width, int height);
    public static native void renderDestroy(long render);
    public static native void renderSetCamera(long render, int projectorType, float[] settings);
    public static native void renderPasses(long render, int[] seeds, int firstBufferSpp);
    public static native void renderRead(long render, float[] out);
    public static native void renderPreview(long render, int[] argbOut);
    /** chunky_render_run: the whole pass loop; postRender is polled from native code. Returns scene.spp. */
    public static native int renderRun(long render, double[] sampleBuffer, int sceneSpp, int targetSpp,
                                       int mergeInterval, java.util.function.BooleanSupplier postRender);

    // tone mapping — replaces the buffers + launch of GpuPostProcessingFilter.java:40-65
    public static native void filterFrame(long ctx, int width, int height, double exposure, double[] input,
                                          int[] argbOut, int type);

    public static final int PALETTE_BLOCK = 0, PALETTE_MATERIAL = 1, PALETTE_AABB = 2, PALETTE_QUAD = 3, PALETTE_TRIG = 4;
    public static final int BVH_WORLD = 0, BVH_ACTOR = 1;
}
