package dev.thatredox.chunkynative.hip;

import org.apache.commons.math3.util.FastMath;
import se.llbit.chunky.renderer.scene.Scene;
import se.llbit.chunky.renderer.scene.Sky;
import se.llbit.chunky.renderer.scene.SkyCache;
import se.llbit.math.Ray;

import java.lang.reflect.Field;

/**
 * The sky bake of ClSky (J/opencl/renderer/scene/ClSky.java:21-76): an equirectangular RGBA8 texture of the sky
 * cache's resolution, i -> theta, j -> phi, bytes (byte) (c * 255), alpha 255, together with the SUN intensity that
 * the kernel multiplies sky samples by (ClSky.java:28-30) — handed to chunky_scene_set_sky.
 *
 * Blind-written (no JDK / chunky-core in the build image); see INTEGRATION.md.
 */
public final class HipSky {
    private HipSky() {}

    public static void upload(long sceneHandle, Scene scene) {
        int res = textureResolution(scene);
        byte[] texture = new byte[res * res * 4];
        Ray ray = new Ray();
        for (int i = 0; i < res; i++) {
            for (int j = 0; j < res; j++) {
                int offset = 4 * (j * res + i);
                double theta = ((double) i / res) * 2 * FastMath.PI;
                double phi = ((double) j / res) * FastMath.PI - FastMath.PI / 2;
                double r = FastMath.cos(phi);
                ray.d.set(FastMath.cos(theta) * r, FastMath.sin(phi), FastMath.sin(theta) * r);
                scene.sky().getSkyColor(ray, false);
                texture[offset] = (byte) (ray.color.x * 255);
                texture[offset + 1] = (byte) (ray.color.y * 255);
                texture[offset + 2] = (byte) (ray.color.z * 255);
                texture[offset + 3] = (byte) 255;
            }
        }
        HipNative.sceneSetSky(sceneHandle, texture, res, res, (float) scene.sun().getIntensity());
    }

    private static int textureResolution(Scene scene) {          // ClSky.java:60-72
        try {
            Sky sky = scene.sky();
            Field skyCacheField = sky.getClass().getDeclaredField("skyCache");
            skyCacheField.setAccessible(true);
            return ((SkyCache) skyCacheField.get(sky)).getSkyResolution();
        } catch (NoSuchFieldException | IllegalAccessException e) {
            throw new RuntimeException(e);
        }
    }
}
