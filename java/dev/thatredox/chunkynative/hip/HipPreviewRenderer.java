package dev.thatredox.chunkynative.hip;

import dev.thatredox.chunkynative.opencl.renderer.scene.ClCamera;   // the reference's class, patched as INTEGRATION.md section 2 says
import se.llbit.chunky.renderer.DefaultRenderManager;
import se.llbit.chunky.renderer.Renderer;
import se.llbit.chunky.renderer.ResetReason;
import se.llbit.chunky.renderer.scene.Scene;

import java.util.function.BooleanSupplier;

/**
 * Drop-in for OpenClPreviewRenderer (J/opencl/OpenClPreviewRenderer.java:25-130): one launch of the
 * `preview` kernel (K/rayTracer.cl:115-217) into the scene's back buffer, then a redraw.
 *
 * Blind-written (no JDK / chunky-core in the build image); see INTEGRATION.md.
 */
public class HipPreviewRenderer implements Renderer {
    private BooleanSupplier postRender = () -> true;
    private final HipSceneLoader sceneLoader;
    private final long ctx;

    public HipPreviewRenderer(long ctx, HipSceneLoader sceneLoader) {
        this.ctx = ctx;
        this.sceneLoader = sceneLoader;
    }

    @Override public String getId() { return "ChunkyClPreviewRenderer"; }               // :26-28
    @Override public String getName() { return "Chunky CL Preview Renderer"; }          // :31-33
    @Override public String getDescription() { return "A work in progress OpenCL renderer."; }
    @Override public void setPostRender(BooleanSupplier callback) { postRender = callback; }
    @Override public boolean autoPostProcess() { return false; }                        // :121-123

    @Override
    public void render(DefaultRenderManager manager) throws InterruptedException {
        Scene scene = manager.bufferedScene;
        int[] imageData = scene.getBackBuffer().data;                                    // :51
        sceneLoader.ensureLoad(scene);                                                   // :54
        long render = HipNative.renderCreate(ctx, sceneLoader.handle(), scene.width, scene.height);
        try {
            ClCamera camera = new ClCamera(scene);                                       // :62
            camera.apply(render);
            camera.generate(render, false);                                              // camera.generate(null, false), :72
            HipNative.renderPreview(render, scene.width, scene.height, imageData);                                 // kernel launch + blocking read, :104-110
            manager.redrawScreen();                                                      // :112
            postRender.getAsBoolean();                                                   // :113
        } finally {
            HipNative.renderDestroy(render);
        }
    }

    @Override
    public void sceneReset(DefaultRenderManager manager, ResetReason reason, int resetCount) {
        sceneLoader.load(resetCount, reason, manager.bufferedScene);                     // :126-128
    }
}
