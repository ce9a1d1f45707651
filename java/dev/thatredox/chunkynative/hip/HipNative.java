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
    /** chunky_group_create: several GPUs behind one context (scenes replicated, the image's 16 x 16 blocks dealt
     *  round-robin, one gather per read-back); every other method takes the handle like init's. */
    public static native long groupCreate(int[] devices);
    public static native int groupSize(long ctx);
    /** chunky_group_peer_status: out[i] = how member i's read-back share reaches member 0 — PEER_LOCAL, PEER_DIRECT (xGMI),
     *  PEER_STAGED (no peer access between the two devices), or a negated HIP error when enabling peer access failed.
     *  out.length must be groupSize(ctx). */
    public static native void groupPeerStatus(long ctx, int[] out);
    /** chunky_group_transport: what carries the one exchange per read-back of a group — TRANSPORT_RCCL_SENDRECV (one grouped
     *  RCCL send / receive of the owned blocks; the default where RCCL could be bound), TRANSPORT_RCCL_REDUCE (one ncclReduce of
     *  the zero-padded framebuffers) or TRANSPORT_PEER_COPY (the fallback) — and, as text, the library / version / ranks or the
     *  reason for the fallback. */
    public static native int groupTransport(long ctx);
    public static native String groupTransportDetail(long ctx);
    /** chunky_group_set_transport; RuntimeException when it needs an RCCL communicator that does not exist. */
    public static native void groupSetTransport(long ctx, int transport);
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
    /** chunky_render_set_option: the render-loop constants of the reference kernel (OPT_DRAW_DEPTH 256, OPT_MAX_DEPTH 5) and
     *  the extensions, all of which default to the reference's behaviour (include/chunky_hip.h). */
    public static native void renderSetOption(long render, int option, int value);
    public static native void renderPasses(long render, int[] seeds, int firstBufferSpp);
    public static native void renderRead(long render, float[] out);
    /** chunky_render_preview; width/height are the render target's, the glue checks argbOut against them. */
    public static native void renderPreview(long render, int width, int height, int[] argbOut);

    /** The hooks of chunky_run_callbacks (include/chunky_hip.h) — what the loop of OpenClPathTracingRenderer.java:95-184
     *  does on the Java side between launches. */
    public interface RunListener {
        /** BooleanSupplier postRender (:153-157,163,181): true stops the loop.  Called for all three polls of the
         *  reference's loop; only the timed one is gated (pollGate). */
        boolean postRender();
        /** Gate of the TIMED poll only: the reference's {@code !manager.shouldFinalize()} (:154).  The polls before a
         *  merge (:163) and after a save event (:181) are unconditional. */
        boolean pollGate();
        /** After every launch: the new scene.spp (:144). */
        void progress(int sceneSpp);
        /** After every merge; the sample buffer is complete (:172-177: postProcessFrame + redrawScreen). */
        void merged(int sampleSpp);
        /** The forced-merge condition of :150 — 1: isSaveEvent(snapshotControl, scene, spp) (:193-195), a snapshot /
         *  dump is due at this spp (merge at once, then one more postRender poll, :179-182); 2: only
         *  scene.shouldFinalizeBuffer() (merge at once, no extra poll); 0: neither. */
        int saveEvent(int spp);
        /** Between launches: re-generate jittered camera rays for non-pinhole projections (:146-148). */
        void regenerateCamera();
    }

    /** chunky_render_run_ex: the whole pass loop.  sampleBuffer is read once and written at every merge (never pinned
     *  across the run).  Returns the new scene.spp. */
    public static native int renderRun(long render, int width, int height, double[] sampleBuffer, int sceneSpp,
                                       int targetSpp, int mergeInterval, RunListener listener);

    // tone mapping — replaces the buffers + launch of GpuPostProcessingFilter.java:40-65
    public static native void filterFrame(long ctx, int width, int height, double exposure, double[] input,
                                          int[] argbOut, int type);

    public static final int PALETTE_BLOCK = 0, PALETTE_MATERIAL = 1, PALETTE_AABB = 2, PALETTE_QUAD = 3, PALETTE_TRIG = 4;
    public static final int BVH_WORLD = 0, BVH_ACTOR = 1;
    public static final int PEER_LOCAL = 0, PEER_DIRECT = 1, PEER_STAGED = 2;
    public static final int TRANSPORT_PEER_COPY = 0, TRANSPORT_RCCL_SENDRECV = 1, TRANSPORT_RCCL_REDUCE = 2;
    public static final int OPT_DRAW_DEPTH = 0, OPT_MAX_DEPTH = 1, OPT_EMITTER_SCALE = 2, OPT_KERNEL = 3, OPT_SUN_SAMPLING = 4,
            OPT_EMITTERS = 5, OPT_BSDF = 6, OPT_EMITTER_NEE = 7, OPT_BVH_CULL_BEHIND = 8;
}
