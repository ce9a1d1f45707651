package dev.thatredox.chunkynative.hip;

import dev.thatredox.chunkynative.util.Reflection;
import dev.thatredox.chunkynative.util.Util;
import se.llbit.chunky.main.Chunky;
import se.llbit.chunky.renderer.scene.Camera;
import se.llbit.chunky.renderer.scene.Scene;
import se.llbit.math.Matrix3;
import se.llbit.math.Ray;
import se.llbit.math.Vector3;

import java.util.Random;
import java.util.concurrent.ThreadLocalRandom;
import java.util.stream.IntStream;

/**
 * ClCamera (J/opencl/renderer/scene/ClCamera.java:26-108) as two array builders: the 15 camera floats of the
 * pinhole projector (position minus the octree origin, the 3x3 transform, aperture, subject distance, fovTan), or
 * width*height*6 pre-generated rays for every other projection mode (projector type -1).
 *
 * Blind-written (no JDK / chunky-core in the build image); see INTEGRATION.md.
 */
public final class HipCamera {
    private HipCamera() {}

    /** True when the projection is not PINHOLE and rays have to be generated on the host (:44-56). */
    public static boolean needGenerate(Scene scene) {
        return scene.camera().getProjectionMode() != se.llbit.chunky.renderer.projection.ProjectionMode.PINHOLE;
    }

    /** chunky_render_set_camera(render, 0, ...): :36-52. */
    public static float[] pinholeSettings(Scene scene) {
        Camera camera = scene.camera();
        Vector3 pos = new Vector3(camera.getPosition());
        pos.sub(scene.getOrigin());
        float[] out = new float[15];
        System.arraycopy(Util.vector3ToFloat(pos), 0, out, 0, 3);
        System.arraycopy(Util.matrix3ToFloat(Reflection.getFieldValue(camera, "transform", Matrix3.class)), 0, out, 3, 9);
        out[12] = camera.infiniteDoF() ? 0 : (float) (camera.getSubjectDistance() / camera.getDof());
        out[13] = (float) camera.getSubjectDistance();
        out[14] = (float) Camera.clampedFovTan(camera.getFov());
        return out;
    }

    /** chunky_render_set_camera(render, -1, ...): the ray table of ClCamera.generate (:72-104). */
    public static float[] generatedRays(Scene scene, boolean jitter) {
        float[] rays = new float[scene.width * scene.height * 3 * 2];
        double halfWidth = scene.width / (2.0 * scene.height);
        double invHeight = 1.0 / scene.height;
        Camera cam = scene.camera();
        Chunky.getCommonThreads().submit(() -> IntStream.range(0, scene.width).parallel().forEach(i -> {
            Ray ray = new Ray();
            Random random = jitter ? ThreadLocalRandom.current() : null;
            for (int j = 0; j < scene.height; j++) {
                int offset = (j * scene.width + i) * 3 * 2;
                float ox = jitter ? random.nextFloat() : 0.5f;
                float oy = jitter ? random.nextFloat() : 0.5f;
                cam.calcViewRay(ray, -halfWidth + (i + ox) * invHeight, -0.5 + (j + oy) * invHeight);
                ray.o.sub(scene.getOrigin());
                System.arraycopy(Util.vector3ToFloat(ray.o), 0, rays, offset, 3);
                System.arraycopy(Util.vector3ToFloat(ray.d), 0, rays, offset + 3, 3);
            }
        })).join();
        return rays;
    }

    /** Sets whichever of the two the scene's projection needs. */
    public static void apply(long render, Scene scene, boolean jitter) {
        if (needGenerate(scene)) HipNative.renderSetCamera(render, -1, generatedRays(scene, jitter));
        else HipNative.renderSetCamera(render, 0, pinholeSettings(scene));
    }
}
