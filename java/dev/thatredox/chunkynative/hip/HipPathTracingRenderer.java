package dev.thatredox.chunkynative.hip;

import dev.thatredox.chunkynative.opencl.renderer.scene.ClCamera;   // the reference's class, patched as INTEGRATION.md section 2 says
import se.llbit.chunky.PersistentSettings;
import se.llbit.chunky.main.Chunky;
import se.llbit.chunky.renderer.DefaultRenderManager;
import se.llbit.chunky.renderer.Renderer;
import se.llbit.chunky.renderer.ResetReason;
import se.llbit.chunky.renderer.SnapshotControl;
import se.llbit.chunky.renderer.scene.Scene;
import se.llbit.util.TaskTracker;

import java.util.concurrent.ForkJoinTask;
import java.util.function.BooleanSupplier;

/**
 * Drop-in for OpenClPathTracingRenderer (J/opencl/OpenClPathTracingRenderer.java): same Renderer
 * ids, same postRender / sceneReset behaviour; the pass loop of :95-184 (seed stream, read-back every
 * <= 1024 passes, double merge) runs inside chunky_render_run_ex and calls back here for everything the
 * reference's loop does on the Java side: scene.spp, post-processing + redraw after a merge, snapshot / dump
 * events, camera-ray regeneration for non-pinhole projections.
 *
 * Blind-written (no JDK / chunky-core in the build image); see INTEGRATION.md.
 */
public class HipPathTracingRenderer implements Renderer {
    private BooleanSupplier postRender = () -> true;
    private final HipSceneLoader sceneLoader;
    private final long ctx;

    public HipPathTracingRenderer(long ctx, HipSceneLoader sceneLoader) {
        this.ctx = ctx;
        this.sceneLoader = sceneLoader;
    }

    @Override public String getId() { return "ChunkyClRenderer"; }          // OpenClPathTracingRenderer.java:33-36
    @Override public String getName() { return "ChunkyClRenderer"; }
    @Override public String getDescription() { return "ChunkyClRenderer"; }
    @Override public void setPostRender(BooleanSupplier callback) { postRender = callback; }
    @Override public boolean autoPostProcess() { return false; }

    @Override
    public void render(DefaultRenderManager manager) throws InterruptedException {
        Scene scene = manager.bufferedScene;
        sceneLoader.ensureLoad(scene);                                       // :64
        long render = HipNative.renderCreate(ctx, sceneLoader.handle(), scene.width, scene.height);
        final ForkJoinTask<?>[] cameraGenTask = {Chunky.getCommonThreads().submit(() -> 0)};   // :97
        try {
            // an extension, off unless the user asks for it (setting "hipBvhCullBehind"): entity-BVH children entirely behind a
            // ray count as missed — about twice the speed on entity-heavy scenes, the reference's image wherever its own
            // arithmetic is meaningful (include/chunky_hip.h CHUNKY_OPT_BVH_CULL_BEHIND)
            if (PersistentSettings.settings.getBool("hipBvhCullBehind", false))
                HipNative.renderSetOption(render, HipNative.OPT_BVH_CULL_BEHIND, 1);
            final ClCamera camera = new ClCamera(scene);                     // :79; patched: ends in HipNative.renderSetCamera
            camera.apply(render);
            camera.generate(render, true);                                   // camera.generate(renderLock, true), :88
            final SnapshotControl snapshots = manager.getSnapshotControl();
            int spp = HipNative.renderRun(render, scene.width, scene.height, scene.getSampleBuffer(), scene.spp,
                    scene.getTargetSpp(), 1024, new HipNative.RunListener() {
                        @Override public boolean postRender() { return postRender.getAsBoolean(); }   // :155,163,181
                        @Override public boolean pollGate() { return !manager.shouldFinalize(); }     // :154, the timed poll only
                        @Override public void progress(int sceneSpp) { scene.spp = sceneSpp; }     // :144
                        @Override public void merged(int sampleSpp) {        // :172-177
                            scene.postProcessFrame(TaskTracker.Task.NONE);
                            manager.redrawScreen();
                        }
                        @Override public int saveEvent(int spp) {            // :150: 1 = isSaveEvent (:193-195), 2 = finalize only
                            if (snapshots.saveSnapshot(scene, spp) || snapshots.saveRenderDump(scene, spp)) return 1;
                            return scene.shouldFinalizeBuffer() ? 2 : 0;
                        }
                        @Override public void regenerateCamera() {           // :146-148 — fresh jitter while passes run; the
                            if (camera.needGenerate && cameraGenTask[0].isDone())   // library's context mutex plays renderLock
                                cameraGenTask[0] = Chunky.getCommonThreads().submit(() -> camera.generate(render, true));
                        }
                    });
            scene.spp = spp;
        } finally {
            cameraGenTask[0].join();                                         // :186
            HipNative.renderDestroy(render);
        }
    }

    @Override
    public void sceneReset(DefaultRenderManager manager, ResetReason reason, int resetCount) {
        sceneLoader.load(resetCount, reason, manager.bufferedScene);         // :203-205
    }
}
