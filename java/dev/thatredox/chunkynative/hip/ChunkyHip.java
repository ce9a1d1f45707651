package dev.thatredox.chunkynative.hip;

import se.llbit.chunky.PersistentSettings;
import se.llbit.chunky.Plugin;
import se.llbit.chunky.main.Chunky;
import se.llbit.chunky.main.ChunkyOptions;
import se.llbit.chunky.renderer.postprocessing.PostProcessingFilters;
import se.llbit.chunky.ui.ChunkyFx;
import se.llbit.log.Log;

/**
 * Plugin entry: what ChunkyCl.attach does (J/opencl/ChunkyCl.java:25-72) with the HIP library in place of the
 * OpenCL device layer — same renderer ids, same imposter filters; the render-controls tab of the OpenCL build
 * (device selection UI) is left to the maintainer.
 *
 * Blind-written (no JDK / chunky-core in the build image); see INTEGRATION.md.
 */
public class ChunkyHip implements Plugin {
    @Override
    public void attach(Chunky chunky) {
        long ctx;
        try {
            // RendererInstance.get(), ChunkyCl.java:35-40.  "hipDevices" = "0" (default) or a list like "0,1,2,3,4,5,6,7":
            // several GPUs behind one context (chunky_group_create) — nothing else in the plugin changes
            String[] ids = PersistentSettings.settings.getString("hipDevices", "0").split(",");
            int[] devices = new int[ids.length];
            for (int i = 0; i < ids.length; i++) devices[i] = Integer.parseInt(ids[i].trim());
            ctx = devices.length == 1 ? HipNative.init(devices[0]) : HipNative.groupCreate(devices);
            if (devices.length > 1) {
                // a member without peer access to member 0 still works (its read-backs are staged through the host): say so
                int[] peers = new int[devices.length];
                HipNative.groupPeerStatus(ctx, peers);
                for (int i = 1; i < peers.length; i++)
                    if (peers[i] != HipNative.PEER_DIRECT && peers[i] != HipNative.PEER_LOCAL)
                        Log.warn("ChunkyHip: GPU " + devices[i] + " has no peer access to GPU " + devices[0] + " (status "
                                + peers[i] + "): its share of every read-back is staged through the host.");
                // one RCCL exchange over xGMI per read-back where the collective library could be bound, peer copies otherwise
                Log.info("ChunkyHip: read-back exchange: " + HipNative.groupTransportDetail(ctx));
            }
        } catch (UnsatisfiedLinkError | RuntimeException e) {
            Log.error("Failed to load ChunkyHip. Could not load libchunky_hip or no gfx950 device.", e);
            return;
        }
        HipSceneLoader sceneLoader = new HipSceneLoader(ctx);
        Chunky.addRenderer(new HipPathTracingRenderer(ctx, sceneLoader));          // :43
        Chunky.addPreviewRenderer(new HipPreviewRenderer(ctx, sceneLoader));       // :44
        addImposterFilter("GAMMA", HipPostProcessingFilter.Filter.GAMMA, ctx);     // :60-63
        addImposterFilter("TONEMAP1", HipPostProcessingFilter.Filter.TONEMAP1, ctx);
        addImposterFilter("TONEMAP2", HipPostProcessingFilter.Filter.ACES, ctx);
        addImposterFilter("TONEMAP3", HipPostProcessingFilter.Filter.HABLE, ctx);
    }

    private static void addImposterFilter(String id, HipPostProcessingFilter.Filter f, long ctx) {   // :66-72
        PostProcessingFilters.getPostProcessingFilterFromId(id).ifPresent(filter ->
                PostProcessingFilters.addPostProcessingFilter(new HipPostProcessingFilter(filter, f, ctx)));
    }

    public static void main(String[] args) throws Exception {                      // :74-80
        Chunky.loadDefaultTextures();
        Chunky chunky = new Chunky(ChunkyOptions.getDefaults());
        new ChunkyHip().attach(chunky);
        ChunkyFx.startChunkyUI(chunky);
    }
}
