package dev.thatredox.chunkynative.hip;

import dev.thatredox.chunkynative.common.export.Packer;
import dev.thatredox.chunkynative.common.export.ResourcePalette;
import it.unimi.dsi.fastutil.ints.IntArrayList;

/**
 * The ClPackedResourcePalette of the HIP build (J/opencl/renderer/export/ClPackedResourcePalette.java:9-37):
 * resources are packed into one int list, `put` returns the offset of the resource in it (the "pointer" the
 * kernels use), and `upload` hands the list to chunky_scene_set_palette — the native side copies, so nothing
 * has to be released.
 *
 * Blind-written (no JDK / chunky-core in the build image); see INTEGRATION.md.
 */
public class HipPalette<T extends Packer> implements ResourcePalette<T> {
    private final long scene;
    private final int kind;
    private final IntArrayList palette = new IntArrayList();
    private boolean uploaded = false;

    public HipPalette(long scene, int kind) {
        this.scene = scene;
        this.kind = kind;
    }

    @Override
    public int put(T resource) {
        if (uploaded) throw new IllegalStateException("Attempted to modify a locked palette.");   // :15
        int ptr = palette.size();
        palette.addAll(resource.pack());
        return ptr;
    }

    /** ClPackedResourcePalette.build(): an empty palette becomes one dummy int natively (ClIntBuffer.java:15-18). */
    public void upload() {
        if (!uploaded) {
            HipNative.sceneSetPalette(scene, kind, palette.toIntArray());
            uploaded = true;
        }
    }
}
