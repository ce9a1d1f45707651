package dev.thatredox.chunkynative.hip;

import dev.thatredox.chunkynative.util.Util;
import se.llbit.chunky.renderer.DefaultRenderManager;
import se.llbit.chunky.renderer.Renderer;
import se.llbit.chunky.renderer.ResetReason;
import se.llbit.chunky.renderer.scene.Camera;
import se.llbit.chunky.renderer.scene.Scene;

import java.util.function.BooleanSupplier;

/**
 * Drop-in for OpenClPathTracingRenderer (J/opencl/OpenClPathTracingRenderer.java): same Renderer
 * ids, same postRender / sceneReset behaviour; the pass loop of :95-184 (seed stream, read-back every
 * <= 1024 passes, double merge) runs inside chunky_render_run.
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
        try {
            HipCamera.apply(render, scene, true);                                    // ClCamera.java:33-104
            int spp = HipNative.renderRun(render, scene.getSampleBuffer(), scene.spp, scene.getTargetSpp(), 1024,
                    () -> {
                        scene.postProcessFrame(se.llbit.util.TaskTracker.Task.NONE);  // :175-176
                        manager.redrawScreen();
                        return postRender.getAsBoolean();
                    });
            scene.spp = spp;
        } finally {
            HipNative.renderDestroy(render);
        }
    }

    @Override
    public void sceneReset(DefaultRenderManager manager, ResetReason reason, int resetCount) {
        sceneLoader.load(resetCount, reason, manager.bufferedScene);         // :203-205
    }
}
