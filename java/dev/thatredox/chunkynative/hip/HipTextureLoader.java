package dev.thatredox.chunkynative.hip;

import dev.thatredox.chunkynative.common.export.texture.AbstractTextureLoader;
import dev.thatredox.chunkynative.common.export.texture.TextureRecord;
import it.unimi.dsi.fastutil.objects.Object2ObjectMap;
import se.llbit.chunky.resources.Texture;

import java.util.ArrayList;
import java.util.Arrays;
import java.util.List;
import java.util.stream.Collectors;

/**
 * The ClTextureLoader of the HIP build (J/opencl/renderer/export/ClTextureLoader.java:19-190): the same packing of
 * textures into 16-texel tiles of 8192x8192 layers (largest first, first free position), the same `location` /
 * `size` record — then one chunky_scene_set_atlas sized to the occupied part and one chunky_scene_write_atlas_tile
 * per texture in place of clCreateImage + clEnqueueWriteImage.
 *
 * Blind-written (no JDK / chunky-core in the build image); see INTEGRATION.md.
 */
public class HipTextureLoader extends AbstractTextureLoader {
    private final long scene;

    public HipTextureLoader(long scene) {
        this.scene = scene;
    }

    @Override
    protected void buildTextures(Object2ObjectMap<Texture, TextureRecord> textures) {
        List<AtlasTexture> texs = textures.entrySet().stream()
                .map(entry -> new AtlasTexture(entry.getKey(), entry.getValue()))
                .sorted().collect(Collectors.toList());                                   // :33-35

        ArrayList<boolean[][]> layers = new ArrayList<>();
        layers.add(new boolean[256][256]);
        for (AtlasTexture tex : texs) {                                                   // :37-44
            if (!insertTex(layers, tex)) {
                layers.add(new boolean[256][256]);
                insertTex(layers, tex);
            }
        }

        // the reference allocates 8192 x 8192 x layers; only the bounding box of the occupied tiles is needed
        // (reads are clamped to the image like CLK_ADDRESS_CLAMP_TO_EDGE would, and every location is inside it)
        int w = 16, h = 16;
        for (AtlasTexture tex : texs) {
            w = Math.max(w, tex.getX() * 16 + tex.getWidth());
            h = Math.max(h, tex.getY() * 16 + tex.getHeight());
        }
        HipNative.sceneSetAtlas(scene, w, h, layers.size());
        for (AtlasTexture tex : texs) {                                                   // :60-67
            HipNative.sceneWriteAtlasTile(scene, tex.getX() * 16, tex.getY() * 16, tex.getD(),
                    tex.getWidth(), tex.getHeight(), tex.getTexture());
        }
        texs.forEach(AtlasTexture::commit);                                               // :69
    }

    private static boolean insertTex(ArrayList<boolean[][]> layers, AtlasTexture tex) {   // :72-86
        int l = 0;
        for (boolean[][] layer : layers) {
            for (int x = 0; x < 256; x++) {
                for (int y = 0; y < 256; y++) {
                    if (insertAt(x, y, tex.getWidth() / 16, tex.getHeight() / 16, layer)) {
                        tex.setLocation(x, y, l);
                        return true;
                    }
                }
            }
            l++;
        }
        return false;
    }

    private static boolean insertAt(int x, int y, int width, int height, boolean[][] layer) {   // :88-114
        if (y + height > layer.length || x + width > layer[0].length) return false;
        if (y < 0 || x < 0) return false;
        for (int line = y; line < y + height; line++)
            for (int pixel = x; pixel < x + width; pixel++)
                if (layer[line][pixel]) return false;
        for (int line = y; line < y + height; line++)
            for (int pixel = x; pixel < x + width; pixel++)
                layer[line][pixel] = true;
        return true;
    }

    /** ClTextureLoader.AtlasTexture (:116-189): size = w << 16 | h, location = x << 22 | y << 13 | layer. */
    protected static class AtlasTexture implements Comparable<AtlasTexture> {
        public final Texture texture;
        public final TextureRecord record;
        public final int size;
        public int location = 0xFFFFFFFF;

        protected AtlasTexture(Texture tex, TextureRecord record) {
            this.texture = tex;
            this.record = record;
            this.size = (tex.getWidth() << 16) | tex.getHeight();
        }

        public void commit() { this.record.set(((long) size << 32) | location); }
        public void setLocation(int x, int y, int d) { this.location = (x << 22) | (y << 13) | d; }
        public int getWidth() { return (size >>> 16) & 0xFFFF; }
        public int getHeight() { return size & 0xFFFF; }
        public int getX() { return (location >>> 22) & 0x1FF; }
        public int getY() { return (location >>> 13) & 0x1FF; }
        public int getD() { return location & 0x1FFF; }

        /** RGBA8 bytes, truncating conversion (byte) (c * 255.0) as the reference (:159-163). */
        public byte[] getTexture() {
            byte[] out = new byte[getHeight() * getWidth() * 4];
            int index = 0;
            for (int y = 0; y < getHeight(); y++) {
                for (int x = 0; x < getWidth(); x++) {
                    float[] rgba = texture.getColor(x, y);
                    out[index] = (byte) (rgba[0] * 255.0);
                    out[index + 1] = (byte) (rgba[1] * 255.0);
                    out[index + 2] = (byte) (rgba[2] * 255.0);
                    out[index + 3] = (byte) (rgba[3] * 255.0);
                    index += 4;
                }
            }
            return out;
        }

        @Override public int compareTo(AtlasTexture o) { return o.size - this.size; }
        @Override public int hashCode() { return Arrays.hashCode(texture.getData()); }

        @Override
        public boolean equals(Object o) {
            if (!(o instanceof AtlasTexture)) return false;
            AtlasTexture other = (AtlasTexture) o;
            return this.size == other.size && Arrays.equals(this.texture.getData(), other.texture.getData());
        }
    }
}
