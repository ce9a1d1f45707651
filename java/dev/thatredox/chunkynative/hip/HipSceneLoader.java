package dev.thatredox.chunkynative.hip;

import dev.thatredox.chunkynative.common.export.AbstractSceneLoader;
import dev.thatredox.chunkynative.common.export.ResourcePalette;
import dev.thatredox.chunkynative.common.export.models.PackedAabbModel;
import dev.thatredox.chunkynative.common.export.models.PackedBvhNode;
import dev.thatredox.chunkynative.common.export.models.PackedQuadModel;
import dev.thatredox.chunkynative.common.export.models.PackedTriangleModel;
import dev.thatredox.chunkynative.common.export.primitives.PackedBlock;
import dev.thatredox.chunkynative.common.export.primitives.PackedMaterial;
import dev.thatredox.chunkynative.common.export.primitives.PackedSun;
import dev.thatredox.chunkynative.common.export.texture.AbstractTextureLoader;
import dev.thatredox.chunkynative.common.state.SkyState;
import dev.thatredox.chunkynative.opencl.renderer.export.ClTextureLoader;   // the reference's classes, patched as
import dev.thatredox.chunkynative.opencl.renderer.scene.ClSky;              // INTEGRATION.md section 2 says
import se.llbit.chunky.renderer.ResetReason;
import se.llbit.chunky.renderer.scene.Scene;

/**
 * The ClSceneLoader of the HIP build (J/opencl/renderer/ClSceneLoader.java): same abstract hooks of
 * AbstractSceneLoader (AbstractSceneLoader.java:184-191), but every palette ends in a
 * chunky_scene_* call instead of a cl_mem.  The packers (PackedBlock, PackedMaterial, ...) and the
 * whole of AbstractSceneLoader.load are reused unchanged — they define the wire formats.
 *
 * Blind-written (no JDK / chunky-core in the build image); see INTEGRATION.md.
 */
public class HipSceneLoader extends AbstractSceneLoader {
    private final long ctx;
    private long scene;
    private boolean skyLoaded = false;
    private SkyState skyState = null;                                    // ClSceneLoader.java:28

    public HipSceneLoader(long ctx) {
        this.ctx = ctx;
        this.scene = HipNative.sceneCreate(ctx);
    }

    public long handle() { return scene; }

    @Override
    public boolean ensureLoad(Scene sceneObj) {
        return this.ensureLoad(sceneObj, !skyLoaded);                       // ClSceneLoader.java:34-36
    }

    @Override
    public boolean load(int modCount, ResetReason resetReason, Scene sceneObj) {
        if (this.modCount != modCount) {                                     // ClSceneLoader.java:40-48: the sky is baked again
            SkyState newSky = new SkyState(sceneObj.sky(), sceneObj.sun());  // only when its state (or the sun's) changed
            if (!newSky.equals(skyState)) {
                new ClSky(scene, sceneObj);   // patched: bakes as before (ClSky.java:41-58), ends in HipNative.sceneSetSky
                skyState = newSky;
                skyLoaded = true;
            }
        }
        Object before = this.blockPalette;
        if (!super.load(modCount, resetReason, sceneObj)) return false;
        if (this.blockPalette != before) {
            // AbstractSceneLoader.load has packed new palettes, BVHs and sun (:93-141): hand them to the library —
            // the OpenCL build does this lazily through ClPackedResourcePalette.get() when kernel arguments are set
            ((HipPalette<PackedBlock>) this.blockPalette).upload();
            ((HipPalette<PackedMaterial>) this.materialPalette.palette).upload();
            ((HipPalette<PackedAabbModel>) this.aabbPalette).upload();
            ((HipPalette<PackedQuadModel>) this.quadPalette).upload();
            ((HipPalette<PackedTriangleModel>) this.trigPalette).upload();
            HipNative.sceneSetBvh(scene, HipNative.BVH_WORLD, this.worldBvh);
            HipNative.sceneSetBvh(scene, HipNative.BVH_ACTOR, this.actorBvh);
            HipNative.sceneSetSun(scene, this.packedSun.pack().toIntArray());
        }
        return true;
    }

    @Override
    protected boolean loadOctree(int[] octree, int depth, int[] blockMapping, ResourcePalette<PackedBlock> blockPalette) {
        // the remap of ClSceneLoader.java:56-58 happens natively (chunky_scene_load_octree)
        HipNative.sceneLoadOctree(scene, octree, depth, blockMapping);
        return true;
    }

    @Override protected AbstractTextureLoader createTextureLoader() { return new ClTextureLoader(scene); }
    @Override protected ResourcePalette<PackedBlock> createBlockPalette() { return new HipPalette<>(scene, HipNative.PALETTE_BLOCK); }
    @Override protected ResourcePalette<PackedMaterial> createMaterialPalette() { return new HipPalette<>(scene, HipNative.PALETTE_MATERIAL); }
    @Override protected ResourcePalette<PackedAabbModel> createAabbModelPalette() { return new HipPalette<>(scene, HipNative.PALETTE_AABB); }
    @Override protected ResourcePalette<PackedQuadModel> createQuadModelPalette() { return new HipPalette<>(scene, HipNative.PALETTE_QUAD); }
    @Override protected ResourcePalette<PackedTriangleModel> createTriangleModelPalette() { return new HipPalette<>(scene, HipNative.PALETTE_TRIG); }

    public void close() {
        if (scene != 0) { HipNative.sceneDestroy(scene); scene = 0; }
    }
}
