"""Synthetic packed scenes in the reference's wire formats.

The reference gets its packed int arrays from Chunky (se.llbit:chunky-core, not available here)
through the packers under `common/export/**`.  This module writes the same int layouts directly:

* octree      — `PackedOctree.treeData` after the leaf remap of `ClSceneLoader.java:52-63`
* block/material/AABB/quad palettes — `PackedBlock.java:80-85`, `PackedMaterial.java:89-100`,
                `PackedAabb.java:96-102` + `PackedAabbModel.java:41-47`, `PackedQuad.java:60-66`
* BVH + triangles — `PackedBvhNode.java:16-31`, `PackedTriangle.java:72-78`
* texture atlas  — tile placement and record packing of `ClTextureLoader.java:72-86,123-131`
* sky / sun / camera — `ClSky.java:23-62`, `PackedSun.java:32-41`, `ClCamera.java:33-70`
* per-pass seeds — `new java.util.Random(0).nextInt()` (`OpenClPathTracingRenderer.java:95,107`)

Everything is seeded numpy; the same call produces the same arrays on every machine, which is
what lets the GPU box regenerate inputs for the committed golden outputs.
"""
from __future__ import annotations

import dataclasses
import math
import struct
from typing import List, Optional, Sequence, Tuple

import numpy as np

ANY_TYPE = 0x7FFFFFFE  # hidden-interior marker, never intersected (block.h:32)
NAN_BITS = 0x7FC00000


def f2i(x: float) -> int:
    """Float.floatToIntBits as a signed int32."""
    return struct.unpack("<i", struct.pack("<f", float(x)))[0]


def java_random_ints(n: int, seed: int = 0) -> np.ndarray:
    """First n values of `new java.util.Random(seed).nextInt()` (48-bit LCG, top 32 bits)."""
    mask = (1 << 48) - 1
    s = (seed ^ 0x5DEECE66D) & mask
    out = np.empty(n, dtype=np.int32)
    for i in range(n):
        s = (s * 0x5DEECE66D + 0xB) & mask
        v = s >> 16
        if v >= 1 << 31:
            v -= 1 << 32
        out[i] = v
    return out


# ------------------------------------------------------------------------------------ textures
class AtlasBuilder:
    """Tile allocator with the reference's placement rule: textures sorted by descending
    size-int, first fit scanning layer, then x (outer), then y (inner) in 16-px tiles
    (`ClTextureLoader.java:32-47,72-86`).  `tiles_x`/`tiles_y` bound the atlas so it stays small;
    the reference uses 256x256 tiles inside an 8192^2 image (same location encoding)."""

    def __init__(self, tiles_x: int = 8, tiles_y: int = 8):
        self.tiles_x, self.tiles_y = tiles_x, tiles_y
        self._tex: List[np.ndarray] = []

    def add(self, rgba: np.ndarray) -> int:
        """rgba: uint8 [h, w, 4]; returns a texture id."""
        assert rgba.dtype == np.uint8 and rgba.shape[2] == 4
        assert rgba.shape[0] % 16 == 0 and rgba.shape[1] % 16 == 0
        self._tex.append(rgba)
        return len(self._tex) - 1

    def build(self) -> Tuple[np.ndarray, List[Tuple[int, int]]]:
        """Returns (atlas uint8 [layers, H, W, 4], [(size_int, location_int)] per texture id)."""
        order = sorted(range(len(self._tex)),
                       key=lambda i: -((self._tex[i].shape[1] << 16) | self._tex[i].shape[0]))
        layers: List[np.ndarray] = [np.zeros((self.tiles_x, self.tiles_y), bool)]
        place = {}
        for i in order:
            h, w = self._tex[i].shape[:2]
            tw, th = w // 16, h // 16
            done = False
            while not done:
                for l, occ in enumerate(layers):
                    for x in range(self.tiles_x - tw + 1):
                        for y in range(self.tiles_y - th + 1):
                            if not occ[x:x + tw, y:y + th].any():
                                occ[x:x + tw, y:y + th] = True
                                place[i] = (x, y, l)
                                done = True
                                break
                        if done:
                            break
                    if done:
                        break
                if not done:
                    layers.append(np.zeros((self.tiles_x, self.tiles_y), bool))
        atlas = np.zeros((len(layers), self.tiles_y * 16, self.tiles_x * 16, 4), np.uint8)
        recs = []
        for i, t in enumerate(self._tex):
            x, y, l = place[i]
            h, w = t.shape[:2]
            atlas[l, y * 16:y * 16 + h, x * 16:x * 16 + w] = t
            recs.append(((w << 16) | h, (x << 22) | (y << 13) | l))
        return atlas, recs


def noise_texture(rng: np.random.Generator, base_rgb: Sequence[int], spread: int = 24,
                  holes: float = 0.0, size: int = 16) -> np.ndarray:
    """A size x size RGBA8 tile: base colour + per-texel noise; `holes` = fraction of fully
    transparent texels (alpha 0 -> Material_sample rejects the hit, material.h:50-54)."""
    t = np.empty((size, size, 4), np.uint8)
    n = rng.integers(-spread, spread + 1, size=(size, size, 1))
    t[..., :3] = np.clip(np.asarray(base_rgb, np.int64)[None, None, :] + n, 0, 255)
    t[..., 3] = 255
    if holes > 0:
        t[rng.random((size, size)) < holes, 3] = 0
    return t


# ------------------------------------------------------------------------------------ palettes
class Palettes:
    """Block / material / AABB-model / quad-model palettes as flat int lists.
    Pointers are int offsets into the respective array, exactly as `ResourcePalette.put`
    hands them out; block k sits at block-palette offset 2k (`PackedBlock.java:80-85`)."""

    def __init__(self):
        self.blocks: List[int] = []
        self.materials: List[int] = []
        self.aabbs: List[int] = []
        self.quads: List[int] = []

    def material(self, *, texture: Optional[Tuple[int, int]] = None, argb: int = 0xFFFFFFFF,
                 tint: int = 0, emittance: float = 0.0, spec: int = 0) -> int:
        ptr = len(self.materials)
        if texture is not None:
            size, loc = texture
            self.materials += [4, tint, size, loc, int(emittance * 255.0), spec]
        else:
            self.materials += [0, tint, 0, argb, int(emittance * 255.0), spec]
        return ptr

    def block_invisible(self) -> int:
        k = len(self.blocks) // 2
        self.blocks += [0, 0]
        return k

    def block_cube(self, material_ptr: int) -> int:
        k = len(self.blocks) // 2
        self.blocks += [1, material_ptr]
        return k

    def block_aabbs(self, boxes: Sequence[Tuple[Sequence[float], int, Sequence[int]]]) -> int:
        """boxes: [(xmin,xmax,ymin,ymax,zmin,zmax), flags, (mN,mE,mS,mW,mT,mB)]"""
        ptr = len(self.aabbs)
        self.aabbs.append(len(boxes))
        for bounds, flags, mats in boxes:
            self.aabbs += [f2i(b) for b in bounds] + [flags] + list(mats)
        k = len(self.blocks) // 2
        self.blocks += [2, ptr]
        return k

    def block_quads(self, quads: Sequence[Tuple[Sequence[float], Sequence[float], Sequence[float],
                                                Sequence[float], int, int]]) -> int:
        """quads: [(o[3], xv[3], yv[3], uv[4]=(u0,du,v0,dv), material, flags)]"""
        ptr = len(self.quads)
        self.quads.append(len(quads))
        for o, xv, yv, uv, mat, flags in quads:
            self.quads += [f2i(v) for v in (*o, *xv, *yv, *uv)] + [mat, flags]
        k = len(self.blocks) // 2
        self.blocks += [3, ptr]
        return k

    @staticmethod
    def _arr(lst: List[int]) -> np.ndarray:
        a = np.array([v if v < (1 << 31) else v - (1 << 32) for v in lst] or [0], dtype=np.int64)
        return a.astype(np.int32)

    def arrays(self):
        return (self._arr(self.blocks), self._arr(self.materials), self._arr(self.aabbs),
                self._arr(self.quads))


# -------------------------------------------------------------------------------------- octree
def build_octree(types: np.ndarray, depth: int, order: str = "bfs") -> np.ndarray:
    """Pack a dense [S,S,S] (S = 2**depth, indexed [x,y,z]) array of block-palette indices
    (or ANY_TYPE) into the reference's `octreeData` layout: `[0]` = root; value v > 0 = index of
    an 8-int child group, child slot (x<<2)|(y<<1)|z; v <= 0 = leaf holding -(2*index) i.e. minus
    the block-palette pointer, ANY_TYPE kept as -0x7FFFFFFE (`ClSceneLoader.java:52-63`,
    `octree.h:81-89`).  Eight equal leaves merge into their parent (Chunky's PackedOctree does
    the same).  Groups are numbered breadth-first ('bfs') or depth-first pre-order ('dfs')."""
    S = 1 << depth
    assert types.shape == (S, S, S)
    BR = -1
    levels = [types.astype(np.int32)]
    for _ in range(depth):
        c = levels[-1]
        h = c.shape[0] // 2
        v = c.reshape(h, 2, h, 2, h, 2).transpose(0, 2, 4, 1, 3, 5).reshape(h, h, h, 8)
        first = v[..., 0]
        same = (v == first[..., None]).all(axis=-1) & (first != BR)
        levels.append(np.where(same, first, BR).astype(np.int32))
    # levels[depth] is the 1x1x1 root

    def leaf_value(t: np.ndarray) -> np.ndarray:
        t = t.astype(np.int64)
        return np.where(t == ANY_TYPE, -ANY_TYPE, -2 * t)

    root = int(levels[depth][0, 0, 0])
    if root != BR:
        return np.array([int(leaf_value(np.array([root]))[0])], np.int32)

    if order == "bfs":
        n_branch = sum(int((l == BR).sum()) for l in levels[1:])
        data = np.zeros(1 + 8 * n_branch, np.int64)
        data[0] = 1
        coords = np.zeros((1, 3), np.int64)  # branch nodes of the current level, in group order
        bases = np.array([1], np.int64)
        nxt = 9
        offs = np.array([[(s >> 2) & 1, (s >> 1) & 1, s & 1] for s in range(8)], np.int64)
        for lvl in range(depth, 0, -1):
            child = levels[lvl - 1]
            cc = (coords[:, None, :] * 2 + offs[None, :, :]).reshape(-1, 3)
            vals = child[cc[:, 0], cc[:, 1], cc[:, 2]].astype(np.int64)
            slots = (bases[:, None] + np.arange(8)[None, :]).reshape(-1)
            isb = vals == BR
            nb = int(isb.sum())
            newbases = nxt + 8 * np.arange(nb, dtype=np.int64)
            out = leaf_value(vals)
            out[isb] = newbases
            data[slots] = out
            coords, bases = cc[isb], newbases
            nxt += 8 * nb
            if nb == 0:
                break
        return data.astype(np.int32)

    # depth-first pre-order, the order Chunky's own loader allocates groups in
    data: List[int] = [0]

    def emit(lvl: int, x: int, y: int, z: int, slot: int) -> None:
        v = int(levels[lvl][x, y, z])
        if v != BR:
            data[slot] = -ANY_TYPE if v == ANY_TYPE else -2 * v
            return
        base = len(data)
        data[slot] = base
        data.extend([0] * 8)
        for s in range(8):
            emit(lvl - 1, 2 * x + ((s >> 2) & 1), 2 * y + ((s >> 1) & 1), 2 * z + (s & 1), base + s)

    import sys
    sys.setrecursionlimit(10000)
    emit(depth, 0, 0, 0, 0)
    return np.array(data, np.int64).astype(np.int32)


def hide_interior(types: np.ndarray, opaque: np.ndarray) -> np.ndarray:
    """Replace cells whose six neighbours are all opaque full cubes by ANY_TYPE, as Chunky does
    for hidden interior (312 369 such leaves in the benchmark octree).
    `opaque[k]` says whether palette index k is an opaque full cube."""
    op = opaque[types]
    inner = op.copy()
    for ax in range(3):
        for sh in (1, -1):
            r = np.roll(op, sh, axis=ax)
            # cells on the boundary have no neighbour on that side: treat as not hidden
            sl = [slice(None)] * 3
            sl[ax] = 0 if sh == 1 else -1
            r[tuple(sl)] = False
            inner &= r
    out = types.copy()
    out[inner] = ANY_TYPE
    return out


# ----------------------------------------------------------------------------------------- BVH
def pack_triangle(v0, v1, v2, uv0, uv1, uv2, material: int, double_sided: bool = False) -> List[int]:
    """20 ints (`PackedTriangle.java:72-78`; Chunky's TexturedTriangle: o = v0... stored as
    e1 = v1-v0?  The kernel only needs o, e1, e2, n consistent with Moller-Trumbore:
    hit = o + u*e1 + v*e2, uv = t1*u + t2*v + t3*(1-u-v) (`primitives.h:368-409`)."""
    v0, v1, v2 = (np.asarray(v, np.float32) for v in (v0, v1, v2))
    e1 = (v1 - v0).astype(np.float32)
    e2 = (v2 - v0).astype(np.float32)
    n = np.cross(e2.astype(np.float64), e1.astype(np.float64))
    # single-sided triangles are front-facing when det = e1.(d x e2) < -EPS; with n = e2 x e1
    # the visible side is the one n points to.
    n = (n / max(np.linalg.norm(n), 1e-30)).astype(np.float32)
    flags = 1 | ((1 << 8) if double_sided else 0)
    vals = [*e1, *e2, *v0, *n, uv1[0], uv1[1], uv2[0], uv2[1], uv0[0], uv0[1]]
    return [flags] + [f2i(v) for v in vals] + [material]


def build_bvh(tris: np.ndarray, leaf_size: int = 4) -> Tuple[np.ndarray, np.ndarray]:
    """Median-split binary BVH over packed triangles `tris` (int32 [n, 20]) in the reference's
    node format (`PackedBvhNode.java:16-31`, walked by `bvh.h:47-109`): 7 ints per node,
    `[0]` > 0 = int index of the second child (first child follows at +7), `[0]` <= 0 = leaf with
    -(pointer into the triangle array); `[1..6]` = xmin,xmax,ymin,ymax,zmin,zmax.  Triangle array:
    per leaf `count` then count x 20 ints.  Returns (nodes int32, trigs int32)."""
    n = tris.shape[0]
    if n == 0:
        return empty_bvh(), np.zeros(1, np.int32)
    f = tris[:, 1:13].copy().view(np.float32).reshape(n, 4, 3)
    e1, e2, o = f[:, 0], f[:, 1], f[:, 2]
    verts = np.stack([o, o + e1, o + e2], axis=1)
    lo, hi = verts.min(axis=1), verts.max(axis=1)
    cent = (lo + hi) * 0.5
    nodes: List[int] = []
    trigs: List[np.ndarray] = []
    tptr = [0]

    def bounds(idx):
        a, b = lo[idx].min(axis=0), hi[idx].max(axis=0)
        return [f2i(a[0]), f2i(b[0]), f2i(a[1]), f2i(b[1]), f2i(a[2]), f2i(b[2])]

    import sys
    sys.setrecursionlimit(100000)

    def rec(idx: np.ndarray) -> None:
        me = len(nodes)
        nodes.extend([0] + bounds(idx))
        if len(idx) <= leaf_size:
            nodes[me] = -tptr[0]
            trigs.append(np.concatenate([[len(idx)], tris[idx].reshape(-1)]).astype(np.int32))
            tptr[0] += 1 + 20 * len(idx)
            return
        ext = cent[idx].max(axis=0) - cent[idx].min(axis=0)
        ax = int(np.argmax(ext))
        srt = idx[np.argsort(cent[idx, ax], kind="stable")]
        mid = len(srt) // 2
        rec(srt[:mid])
        nodes[me] = len(nodes)
        rec(srt[mid:])

    rec(np.arange(n))
    return np.array(nodes, np.int64).astype(np.int32), np.concatenate(trigs).astype(np.int32)


def empty_bvh() -> np.ndarray:
    """`PackedBvhNode.java:16-18`: {0, NaN x 6}."""
    return np.array([0] + [NAN_BITS] * 6, np.int32)


# ------------------------------------------------------------------------------ sky, sun, camera
def bake_sky(res: int = 128, sun_dir: Optional[Sequence[float]] = None) -> np.ndarray:
    """res x res RGBA8 equirect bake: texel (i, j) holds the colour of direction
    theta = 2*pi*i/res, phi = pi*j/res - pi/2, d = (cos t cos p, sin p, sin t cos p), bytes
    (byte)(c*255), alpha 255 (`ClSky.java:43-58`).  Colour = analytic gradient (+ glow)."""
    i = np.arange(res, dtype=np.float64)
    th = 2 * np.pi * i / res
    ph = np.pi * i / res - np.pi / 2
    TH, PH = np.meshgrid(th, ph, indexing="xy")  # [j, i]
    d = np.stack([np.cos(TH) * np.cos(PH), np.sin(PH), np.sin(TH) * np.cos(PH)], axis=-1)
    up = np.clip(d[..., 1], -1, 1)
    horizon = np.array([0.80, 0.88, 0.97])
    zenith = np.array([0.25, 0.45, 0.85])
    ground = np.array([0.32, 0.30, 0.28])
    t = np.clip(up, 0, 1)[..., None] ** 0.5
    col = horizon * (1 - t) + zenith * t
    g = np.clip(-up, 0, 1)[..., None] ** 0.3
    col = np.where(up[..., None] < 0, horizon * (1 - g) + ground * g, col)
    if sun_dir is not None:
        s = np.asarray(sun_dir, np.float64)
        glow = np.clip((d @ s), 0, 1) ** 32
        col = np.clip(col + 0.25 * glow[..., None], 0, 1)
    out = np.empty((res, res, 4), np.uint8)
    out[..., :3] = (col * 255.0).astype(np.int64).astype(np.uint8)
    out[..., 3] = 255
    return out


def pack_sun(altitude: float, azimuth: float, intensity: float, draw_texture: bool,
             texture: Tuple[int, int] = (0, 0)) -> np.ndarray:
    """`PackedSun.java:32-41`: flags, textureSize, textureLocation, intensity, altitude, azimuth."""
    return np.array([1 if draw_texture else 0, texture[0], texture[1], f2i(intensity),
                     f2i(altitude), f2i(azimuth)], np.int64).astype(np.int32)


def sun_direction(altitude: float, azimuth: float) -> np.ndarray:
    r = abs(math.cos(altitude))
    return np.array([math.cos(azimuth) * r, math.sin(altitude), math.sin(azimuth) * r])


def look_at_camera(pos: Sequence[float], target: Sequence[float], fov_deg: float = 70.0,
                   aperture: float = 0.0, subject_distance: float = 2.0) -> np.ndarray:
    """15 floats of `ClCamera.java:42-52`: pos, 3x3 transform rows (world = M * cam), aperture,
    subjectDistance, fovTan.  Camera space: x right, y down the image, z forward
    (`rayTracer.cl:68-69`: y grows with the pixel row)."""
    p = np.asarray(pos, np.float64)
    fwd = np.asarray(target, np.float64) - p
    fwd /= np.linalg.norm(fwd)
    up = np.array([0.0, 1.0, 0.0])
    right = np.cross(up, fwd)
    right /= np.linalg.norm(right)
    down = -np.cross(fwd, right)
    M = np.stack([right, down, fwd], axis=1)  # columns = camera axes in world space
    # Chunky's Camera.clampedFovTan: 2*tan(fov/2) for the pinhole projector
    fov_tan = 2.0 * math.tan(math.radians(fov_deg) / 2.0)
    return np.concatenate([p, M.reshape(-1), [aperture, subject_distance, fov_tan]]).astype(np.float32)


# --------------------------------------------------------------------------------------- scene
@dataclasses.dataclass
class PackedScene:
    """All device inputs of one render (the 20 kernel arguments of rayTracer.cl:11-37)."""
    octree: np.ndarray
    octree_depth: int
    block_palette: np.ndarray
    material_palette: np.ndarray
    aabb_models: np.ndarray
    quad_models: np.ndarray
    world_bvh: np.ndarray
    actor_bvh: np.ndarray
    bvh_trigs: np.ndarray
    atlas: np.ndarray            # uint8 [layers, H, W, 4]
    sky: np.ndarray              # uint8 [res, res, 4]
    sky_intensity: float
    sun: np.ndarray              # int32 [6]
    camera: np.ndarray           # float32 [15] or [W*H*6]
    projector_type: int = 0
    width: int = 64
    height: int = 64
    name: str = "scene"

    def with_view(self, width: int, height: int, camera: Optional[np.ndarray] = None,
                  projector_type: Optional[int] = None) -> "PackedScene":
        return dataclasses.replace(
            self, width=width, height=height,
            camera=self.camera if camera is None else camera,
            projector_type=self.projector_type if projector_type is None else projector_type)


def value_noise2(rng: np.random.Generator, n: int, octaves: int = 4, base: int = 64) -> np.ndarray:
    """[n, n] value noise in [0, 1): bilinear-interpolated random lattices, `octaves` octaves."""
    out = np.zeros((n, n))
    amp, tot = 1.0, 0.0
    cell = base
    for _ in range(octaves):
        g = max(n // cell, 1) + 2
        lat = rng.random((g, g))
        xs = np.arange(n) / cell
        i0 = np.floor(xs).astype(int)
        f = xs - i0
        f = f * f * (3 - 2 * f)
        a = lat[i0][:, i0]
        b = lat[i0 + 1][:, i0]
        c = lat[i0][:, i0 + 1]
        d = lat[i0 + 1][:, i0 + 1]
        fx, fz = f[:, None], f[None, :]
        out += amp * ((a * (1 - fx) + b * fx) * (1 - fz) + (c * (1 - fx) + d * fx) * fz)
        tot += amp
        amp *= 0.5
        cell = max(cell // 2, 1)
    return out / tot


def outdoor_world(chunks: int = 32, height: int = 256, seed: int = 20260101,
                  width: int = 1920, img_height: int = 1080, sun_flag: bool = True,
                  aabb_frac: float = 0.02, quad_frac: float = 0.01, trees: bool = True,
                  emitters: float = 0.0, octree_order: str = "bfs", atlas_tiles: Tuple[int, int] = (8, 8),
                  water: bool = False) -> PackedScene:
    """BASELINE.json config 3: chunks x chunks Minecraft chunks (16 columns each), `height` tall,
    seeded value-noise terrain with bedrock/stone/dirt/grass layers, ANY_TYPE hidden interior,
    ~`aabb_frac` slab (AABB-model) and ~`quad_frac` plant (quad-model) blocks on the surface,
    optional trees with alpha-cut-out leaf cubes, 12 procedural 16x16 textures, 128^2 baked sky,
    sun altitude 0.6 / azimuth 1.2 / intensity 1.25.  With chunks=32, height=256 the world is
    512x256x512 blocks inside a depth-9 octree.

    `atlas_tiles` bounds the atlas layers (in 16-px tiles): (2, 2) pushes the thirteen textures over four layers, with
    tiles at y = 1 whose location bit 13 runs into the kernel's 19-bit layer mask (K/textureAtlas.h:13, quirk B#7).
    `water` floods the low ground with a cube whose material carries the biome-water tint (type 3, K/material.h:61-72)."""
    rng = np.random.default_rng(seed)
    n = chunks * 16
    S = max(n, height)
    depth = int(math.ceil(math.log2(S)))
    S = 1 << depth

    # --- textures + materials -------------------------------------------------------------
    ab = AtlasBuilder(*atlas_tiles)
    tex = {
        "bedrock": ab.add(noise_texture(rng, (60, 60, 60), 30)),
        "stone": ab.add(noise_texture(rng, (125, 125, 125), 14)),
        "dirt": ab.add(noise_texture(rng, (134, 96, 67), 16)),
        "grass": ab.add(noise_texture(rng, (150, 150, 150), 18)),
        "sand": ab.add(noise_texture(rng, (219, 207, 163), 10)),
        "log": ab.add(noise_texture(rng, (102, 81, 50), 12)),
        "leaves": ab.add(noise_texture(rng, (140, 140, 140), 30, holes=0.25)),
        "plank": ab.add(noise_texture(rng, (162, 130, 78), 10)),
        "plant": ab.add(noise_texture(rng, (120, 160, 120), 30, holes=0.55)),
        "glow": ab.add(noise_texture(rng, (250, 220, 150), 5)),
        "snow": ab.add(noise_texture(rng, (240, 245, 250), 6)),
        "sun": ab.add(noise_texture(rng, (255, 250, 230), 4, size=32)),
    }
    if water:
        tex["water"] = ab.add(noise_texture(rng, (200, 200, 200), 12))
    atlas, recs = ab.build()
    pal = Palettes()
    m = {k: pal.material(texture=recs[v]) for k, v in tex.items() if k not in ("grass", "leaves", "plant", "glow", "sun", "water")}
    m["grass"] = pal.material(texture=recs[tex["grass"]], tint=2 << 24)
    m["leaves"] = pal.material(texture=recs[tex["leaves"]], tint=1 << 24)
    m["plant"] = pal.material(texture=recs[tex["plant"]], tint=0xFF000000 | 0x6FB040)
    m["glow"] = pal.material(texture=recs[tex["glow"]], emittance=1.0 if emitters > 0 else 0.0)
    m["flat"] = pal.material(argb=0xFF3060C0)
    if water:
        m["water"] = pal.material(texture=recs[tex["water"]], tint=3 << 24)

    B = {"air": pal.block_invisible()}
    for k in ("bedrock", "stone", "dirt", "grass", "sand", "log", "leaves", "plank", "glow", "snow", "flat"):
        B[k] = pal.block_cube(m[k])
    # bottom slab: one box, all faces present, side faces rotated differently to exercise flags
    slab_flags = 0 | (0b0110 << 4) | (0b0000 << 8) | (0b0101 << 12) | (0b0011 << 16) | (0b0100 << 20)
    B["slab"] = pal.block_aabbs([((0, 1, 0, 0.5, 0, 1), slab_flags,
                                  (m["plank"], m["plank"], m["plank"], m["plank"], m["plank"], m["plank"]))])
    # fence-like post + rail: two boxes, bottom face of the rail absent
    B["post"] = pal.block_aabbs([
        ((0.375, 0.625, 0, 1, 0.375, 0.625), 0, (m["log"],) * 6),
        ((0, 1, 0.375, 0.5625, 0.4375, 0.5625), 0b1000 << 20, (m["plank"],) * 6)])
    # crossed-quads plant (two quads per diagonal so it is visible from both sides; the kernel
    # ignores the doubleSided flag, primitives.h:298-319)
    def quad(o, xv, yv):
        return (o, xv, yv, (0.0, 1.0, 0.0, 1.0), m["plant"], 1)
    B["plant"] = pal.block_quads([
        quad((0.15, 0, 0.15), (0.7, 0, 0.7), (0, 1, 0)),
        quad((0.85, 0, 0.85), (-0.7, 0, -0.7), (0, 1, 0)),
        quad((0.15, 0, 0.85), (0.7, 0, -0.7), (0, 1, 0)),
        quad((0.85, 0, 0.15), (-0.7, 0, 0.7), (0, 1, 0))])
    if water:
        B["water"] = pal.block_cube(m["water"])
    nblocks = len(pal.blocks) // 2
    opaque = np.zeros(nblocks, bool)
    for k in ("bedrock", "stone", "dirt", "grass", "sand", "log", "plank", "glow", "snow", "flat"):
        opaque[B[k]] = True

    # --- terrain ---------------------------------------------------------------------------
    hmap = value_noise2(rng, n, octaves=4, base=max(n // 4, 8))
    lo_h, hi_h = 0.19 * height, 0.55 * height
    hgt = (lo_h + (hi_h - lo_h) * hmap).astype(np.int32)          # [x, z] top solid y
    types = np.zeros((S, S, S), np.int32)                           # [x, y, z]
    ys = np.arange(S, dtype=np.int32)[None, :, None]
    H3 = np.zeros((S, 1, S), np.int32)
    H3[:n, 0, :n] = hgt
    inside = np.zeros((S, 1, S), bool)
    inside[:n, 0, :n] = True
    t = types
    t[(ys <= H3) & inside] = B["stone"]
    t[(ys <= H3) & (ys > H3 - 4) & inside] = B["dirt"]
    beach = (H3 < lo_h + 0.08 * (hi_h - lo_h))
    peak = (H3 > lo_h + 0.85 * (hi_h - lo_h))
    t[(ys == H3) & inside] = B["grass"]
    t[(ys <= H3) & (ys > H3 - 3) & inside & beach] = B["sand"]
    t[(ys == H3) & inside & peak] = B["snow"]
    t[(ys == 0) & inside] = B["bedrock"]
    if water:  # a water table a little above the beach line: every column below it is filled up to it
        level = int(lo_h + 0.2 * (hi_h - lo_h))
        t[(ys > H3) & (ys <= level) & inside] = B["water"]

    # surface decorations
    r = rng.random((n, n))
    xs, zs = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
    top = hgt + 1
    ok = top < height - 12

    def place(mask, block):
        sel = mask & ok
        t[xs[sel], top[sel], zs[sel]] = block

    place(r < aabb_frac * 0.6, B["slab"])
    place((r >= aabb_frac * 0.6) & (r < aabb_frac), B["post"])
    place((r >= aabb_frac) & (r < aabb_frac + quad_frac), B["plant"])
    if emitters > 0:
        place((r >= 0.5) & (r < 0.5 + emitters), B["glow"])
    if trees:
        tr = (r >= 0.9) & (r < 0.9035) & ok & (xs > 3) & (xs < n - 4) & (zs > 3) & (zs < n - 4) & ~beach[:n, 0, :n]
        for x, z in zip(xs[tr], zs[tr]):
            y0 = int(top[x, z])
            th = 4 + int(rng.integers(0, 3))
            t[x - 2:x + 3, y0 + th - 2:y0 + th, z - 2:z + 3] = B["leaves"]
            t[x - 1:x + 2, y0 + th:y0 + th + 2, z - 1:z + 2] = B["leaves"]
            t[x, y0:y0 + th, z] = B["log"]

    t = hide_interior(t, opaque)
    octree = build_octree(t, depth, octree_order)
    blocks, mats, aabbs, quads = pal.arrays()

    alt, azi, inten = 0.6, 1.2, 1.25
    sky = bake_sky(128, sun_direction(alt, azi))
    sun = pack_sun(alt, azi, inten, sun_flag, recs[tex["sun"]])
    cx = n / 2
    cam = look_at_camera((n * 0.18, hi_h + 0.20 * height, n * 0.15),
                         (cx, lo_h + 0.3 * (hi_h - lo_h), cx * 1.1), fov_deg=70.0)
    return PackedScene(octree=octree, octree_depth=depth, block_palette=blocks,
                       material_palette=mats, aabb_models=aabbs, quad_models=quads,
                       world_bvh=empty_bvh(), actor_bvh=empty_bvh(), bvh_trigs=np.zeros(1, np.int32),
                       atlas=atlas, sky=sky, sky_intensity=inten, sun=sun, camera=cam,
                       width=width, height=img_height, name=f"outdoor{chunks}x{chunks}")


def indoor_room(size: int = 64, seed: int = 7, width: int = 1920, img_height: int = 1080,
                emitter_frac: float = 0.01) -> PackedScene:
    """BASELINE.json config 4: a closed `size`^3 room inside a depth-ceil(log2(size+2)) octree,
    ~`emitter_frac` of the wall/ceiling cells emissive (emittance byte 255), sun flag 0
    (no sun draws; `PackedSun.java:16`, `sky.h:69`), a few pillars and slabs inside."""
    rng = np.random.default_rng(seed)
    depth = int(math.ceil(math.log2(size + 2)))
    S = 1 << depth
    ab = AtlasBuilder(4, 4)
    tw = ab.add(noise_texture(rng, (200, 200, 195), 10))
    tf = ab.add(noise_texture(rng, (120, 90, 70), 14))
    tl = ab.add(noise_texture(rng, (255, 240, 200), 3))
    tp = ab.add(noise_texture(rng, (90, 110, 160), 12))
    atlas, recs = ab.build()
    pal = Palettes()
    mw, mf = pal.material(texture=recs[tw]), pal.material(texture=recs[tf])
    ml = pal.material(texture=recs[tl], emittance=1.0)
    mp = pal.material(texture=recs[tp])
    air = pal.block_invisible()
    wall, floor, lamp, pillar = (pal.block_cube(x) for x in (mw, mf, ml, mp))
    slab = pal.block_aabbs([((0, 1, 0, 0.5, 0, 1), 0, (mp,) * 6)])
    t = np.zeros((S, S, S), np.int32)
    a, b = 0, size + 1
    t[a:b + 1, a:b + 1, a:b + 1] = wall
    t[a + 1:b, a + 1:b, a + 1:b] = air
    t[a:b + 1, a, a:b + 1] = floor
    shell = np.zeros_like(t, bool)
    shell[a:b + 1, a:b + 1, a:b + 1] = True
    shell[a + 1:b, a + 1:b, a + 1:b] = False
    shell[:, a, :] = False
    lamps = shell & (rng.random(t.shape) < emitter_frac)
    t[lamps] = lamp
    for _ in range(max(size // 8, 2)):
        x, z = rng.integers(a + 4, b - 4, size=2)
        hgt = int(rng.integers(size // 4, size - 2))
        t[x:x + 2, a + 1:a + 1 + hgt, z:z + 2] = pillar
        t[x - 1, a + 1, z] = slab
    octree = build_octree(t, depth, "bfs")
    blocks, mats, aabbs, quads = pal.arrays()
    sky = bake_sky(128)
    sun = pack_sun(0.6, 1.2, 1.25, False)
    c = size / 2 + 1
    cam = look_at_camera((a + 3.5, a + size * 0.6, a + 4.5), (c + size * 0.2, a + size * 0.3, c + size * 0.15), 80.0)
    return PackedScene(octree=octree, octree_depth=depth, block_palette=blocks,
                       material_palette=mats, aabb_models=aabbs, quad_models=quads,
                       world_bvh=empty_bvh(), actor_bvh=empty_bvh(), bvh_trigs=np.zeros(1, np.int32),
                       atlas=atlas, sky=sky, sky_intensity=1.25, sun=sun, camera=cam,
                       width=width, height=img_height, name=f"indoor{size}")


def add_entities(scene: PackedScene, n_tris: int, seed: int = 11, actor_tris: int = 0,
                 region: Optional[Tuple[Sequence[float], Sequence[float]]] = None,
                 leaf_size: int = 4) -> PackedScene:
    """BASELINE.json config 5: seeded instanced meshes (small textured tetrahedra/boxes) scattered
    in `region` (default: upper half of the world cube), packed into the world BVH (and
    optionally an actor BVH) that share one triangle array, as `AbstractSceneLoader.java:118-127`
    does."""
    rng = np.random.default_rng(seed)
    S = float(1 << scene.octree_depth)
    lo, hi = region if region is not None else ((0.1 * S, 0.3 * S, 0.1 * S), (0.6 * S, 0.7 * S, 0.6 * S))
    lo, hi = np.asarray(lo), np.asarray(hi)
    nmat = len(scene.material_palette) // 6

    def mesh(count):
        out = []
        k = 0
        while k < count:
            c = lo + rng.random(3) * (hi - lo)
            s = 0.3 + rng.random() * 1.2
            R = np.linalg.qr(rng.normal(size=(3, 3)))[0]
            v = (np.array([[1, 1, 1], [1, -1, -1], [-1, 1, -1], [-1, -1, 1]], float) * s) @ R.T + c
            mat = 6 * int(rng.integers(0, nmat))
            ds = bool(rng.random() < 0.3)
            for (i, j, l) in ((0, 1, 2), (0, 3, 1), (0, 2, 3), (1, 3, 2)):
                out.append(pack_triangle(v[i], v[j], v[l], (0, 0), (1, 0), (0, 1), mat, ds))
                k += 1
                if k >= count:
                    break
        return np.array(out, np.int64).astype(np.int32).reshape(-1, 20)

    wt = mesh(n_tris)
    wn, wtr = build_bvh(wt, leaf_size)
    if actor_tris > 0:
        at = mesh(actor_tris)
        an, atr = build_bvh(at, leaf_size)
        # shift actor leaf pointers behind the world triangles (one shared triangle array)
        an = an.copy().reshape(-1, 7)
        leaf = an[:, 0] <= 0
        an[leaf, 0] -= len(wtr)
        an = an.reshape(-1)
        trigs = np.concatenate([wtr, atr])
    else:
        an, trigs = empty_bvh(), wtr
    return dataclasses.replace(scene, world_bvh=wn, actor_bvh=an, bvh_trigs=trigs,
                               name=scene.name + f"+{n_tris}tri")


def tiny_scene(seed: int = 3, size: int = 16, width: int = 48, height: int = 32,
               entities: int = 40, sun_flag: bool = True) -> PackedScene:
    """A small world for unit tests: the outdoor generator at 1 chunk, plus a few entities."""
    sc = outdoor_world(chunks=max(size // 16, 1), height=size * 2, seed=seed, width=width,
                       img_height=height, sun_flag=sun_flag, aabb_frac=0.10, quad_frac=0.06,
                       emitters=0.02)
    if entities:
        S = float(1 << sc.octree_depth)
        sc = add_entities(sc, entities, seed=seed + 1, actor_tris=entities // 2,
                          region=((1, 0.35 * S, 1), (size - 1, 0.75 * S, size - 1)))
    return sc


# ---------------------------------------------------------------------------------------- cache
_ARRAY_FIELDS = ("octree", "block_palette", "material_palette", "aabb_models", "quad_models", "world_bvh",
                 "actor_bvh", "bvh_trigs", "atlas", "sky", "sun", "camera")


def save_scene(sc: PackedScene, path: str, compressed: bool = False) -> None:
    import os
    os.makedirs(os.path.dirname(path), exist_ok=True)
    tmp = path + f".{os.getpid()}.tmp.npz"
    (np.savez_compressed if compressed else np.savez)(tmp, meta=np.array([sc.octree_depth, sc.projector_type, sc.width, sc.height], np.int64),
             sky_intensity=np.float64(sc.sky_intensity), name=np.array(sc.name),
             **{f: getattr(sc, f) for f in _ARRAY_FIELDS})
    os.replace(tmp, path)


def save_raw(sc: PackedScene, path: str) -> None:
    """The scene as a flat dump a host without numpy reads (examples/host_example.cpp): "CHKSCN01", then per array a record
    {name[16], int32 dtype (0 int32, 1 uint8, 2 float32), int32 ndim, int64 dims[4], payload}, little-endian."""
    import struct
    arrays = [("meta", np.array([sc.octree_depth, sc.projector_type, sc.width, sc.height], np.int32)),
              ("sky_intensity", np.array([sc.sky_intensity], np.float32))]
    for f in _ARRAY_FIELDS:
        a = getattr(sc, f)
        dt = np.uint8 if f in ("atlas", "sky") else (np.float32 if f == "camera" else np.int32)
        arrays.append((f, np.ascontiguousarray(a, dt)))
    with open(path, "wb") as out:
        out.write(b"CHKSCN01")
        for name, a in arrays:
            code = {np.dtype(np.int32): 0, np.dtype(np.uint8): 1, np.dtype(np.float32): 2}[a.dtype]
            dims = list(a.shape) + [1] * (4 - a.ndim)
            out.write(name.encode().ljust(16, b"\0"))
            out.write(struct.pack("<ii4q", code, a.ndim, *dims))
            out.write(a.tobytes())


def load_scene(path: str) -> PackedScene:
    z = np.load(path)
    depth, proj, w, h = (int(v) for v in z["meta"])
    return PackedScene(octree_depth=depth, projector_type=proj, width=w, height=h,
                       sky_intensity=float(z["sky_intensity"]), name=str(z["name"]),
                       **{f: z[f] for f in _ARRAY_FIELDS})


def cached_outdoor_world(**kw) -> PackedScene:
    """outdoor_world(**kw) through an on-disk cache (.scene_cache/ next to the package): the
    32x32-chunk world takes ~30 s of numpy to generate and is byte-identical every time."""
    import hashlib
    import inspect
    import os
    sig = inspect.signature(outdoor_world)
    bound = sig.bind(**kw)
    bound.apply_defaults()
    src = inspect.getsource(outdoor_world) + inspect.getsource(build_octree) + inspect.getsource(hide_interior)
    key = hashlib.sha256((repr(sorted(bound.arguments.items())) + src).encode()).hexdigest()[:16]
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), ".scene_cache")
    path = os.path.join(root, f"outdoor_{key}.npz")
    if os.path.exists(path):
        try:
            return load_scene(path)
        except Exception:
            pass
    sc = outdoor_world(**kw)
    try:
        save_scene(sc, path)
    except OSError:
        pass
    return sc



# ------------------------------------------------------------------------------- a world beyond the caches
def build_octree_slab(t: np.ndarray, depth: int, any_code: int) -> np.ndarray:
    """build_octree(order="bfs") for a world that is a SLAB: `t` is [n, h, n] (n = 2**depth, h a power of two <= n) of small
    block-palette indices (int16; `any_code` stands for ANY_TYPE), everything above y = h is air.  Same layout, same merging,
    same breadth-first group numbering as build_octree on the dense [n, n, n] array — which for a 2048-wide world would be
    34 GB — built from the slab's own halvings; from the level where the slab is one cell thick it is padded with air."""
    n, h = t.shape[0], t.shape[1]
    assert t.shape == (n, h, n) and n == 1 << depth and h & (h - 1) == 0 and h <= n and t.dtype == np.int16
    BR = -1
    levels = [t]
    for _ in range(depth):
        c = levels[-1]
        if c.shape[1] == 1 and c.shape[0] > 1:  # the slab is one cell thick here: the cells above it are air
            full = np.zeros((c.shape[0],) * 3, np.int16)
            full[:, 0, :] = c[:, 0, :]
            c = full
        a, b = c.shape[0] // 2, c.shape[1] // 2
        v = c.reshape(a, 2, b, 2, a, 2).transpose(0, 2, 4, 1, 3, 5).reshape(a, b, a, 8)
        first = v[..., 0]
        same = (v == first[..., None]).all(axis=-1) & (first != BR)
        levels.append(np.where(same, first, BR).astype(np.int16))

    def leaf_value(x: np.ndarray) -> np.ndarray:
        x = x.astype(np.int64)
        return np.where(x == any_code, -ANY_TYPE, -2 * x)

    def at(level: np.ndarray, cc: np.ndarray) -> np.ndarray:  # cells above the slab are air
        hh = level.shape[1]
        ok = cc[:, 1] < hh
        out = np.zeros(len(cc), np.int64)
        out[ok] = level[cc[ok, 0], cc[ok, 1], cc[ok, 2]]
        return out

    root = int(levels[depth][0, 0, 0])
    if root != BR:
        return np.array([int(leaf_value(np.array([root]))[0])], np.int32)
    n_branch = sum(int((l == BR).sum()) for l in levels[1:])
    data = np.zeros(1 + 8 * n_branch, np.int64)
    data[0] = 1
    coords = np.zeros((1, 3), np.int64)
    bases = np.array([1], np.int64)
    nxt = 9
    offs = np.array([[(s >> 2) & 1, (s >> 1) & 1, s & 1] for s in range(8)], np.int64)
    for lvl in range(depth, 0, -1):
        cc = (coords[:, None, :] * 2 + offs[None, :, :]).reshape(-1, 3)
        vals = at(levels[lvl - 1], cc)
        slots = (bases[:, None] + np.arange(8)[None, :]).reshape(-1)
        isb = vals == BR
        nb = int(isb.sum())
        newbases = nxt + 8 * np.arange(nb, dtype=np.int64)
        out = leaf_value(vals)
        out[isb] = newbases
        data[slots] = out
        coords, bases = cc[isb], newbases
        nxt += 8 * nb
        if nb == 0:
            break
    assert nxt == len(data)
    return data.astype(np.int32)


def big_outdoor_world(chunks: int = 128, height: int = 256, seed: int = 20260606, width: int = 1920, img_height: int = 1080) -> PackedScene:
    """A world whose trees do NOT fit the 256 MiB Infinity Cache (bench.py --config 5; not a BASELINE configuration): chunks x chunks
    chunks of the outdoor world's kind — the same palettes, textures, sky and sun (taken from outdoor_world itself), value-noise
    terrain with the same layers, slabs / posts / plants on the surface, hidden interior as ANY_TYPE — 2048 x 256 x 2048 blocks in a
    depth-11 octree at the default size."""
    small = outdoor_world(chunks=1, height=16, width=width, img_height=img_height)
    names = ("air", "bedrock", "stone", "dirt", "grass", "sand", "log", "leaves", "plank", "glow", "snow", "flat", "slab", "post", "plant")
    B = {k: i for i, k in enumerate(names)}   # outdoor_world's block order (asserted below through the palette's model types)
    assert len(small.block_palette) == 2 * len(names) and list(small.block_palette[2 * B["slab"]::2][:3]) == [2, 2, 3]
    ANY = 255
    rng = np.random.default_rng(seed)
    n = chunks * 16
    depth = int(math.ceil(math.log2(max(n, height))))
    assert n == 1 << depth and height & (height - 1) == 0, "chunks x 16 and the height have to be powers of two"
    hmap = value_noise2(rng, n, octaves=5, base=max(n // 8, 8))
    lo_h, hi_h = 0.19 * height, 0.55 * height
    hgt = (lo_h + (hi_h - lo_h) * hmap).astype(np.int16)
    t = np.zeros((n, height, n), np.int16)
    ys = np.arange(height, dtype=np.int16)[None, :, None]
    H3 = hgt[:, None, :]
    t[ys <= H3] = B["stone"]
    t[(ys <= H3) & (ys > H3 - 4)] = B["dirt"]
    beach = H3 < lo_h + 0.08 * (hi_h - lo_h)
    peak = H3 > lo_h + 0.85 * (hi_h - lo_h)
    t[ys == H3] = B["grass"]
    t[(ys <= H3) & (ys > H3 - 3) & beach] = B["sand"]
    t[(ys == H3) & peak] = B["snow"]
    t[:, 0, :] = B["bedrock"]
    r = rng.random((n, n))
    xs, zs = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
    top = hgt.astype(np.int64) + 1

    def place(mask, block):
        t[xs[mask], top[mask], zs[mask]] = block

    place(r < 0.012, B["slab"])
    place((r >= 0.012) & (r < 0.02), B["post"])
    place((r >= 0.02) & (r < 0.03), B["plant"])
    tr = (r >= 0.9) & (r < 0.9035) & (xs > 3) & (xs < n - 4) & (zs > 3) & (zs < n - 4) & ~beach[:, 0, :]
    for x, z in zip(xs[tr], zs[tr]):
        y0 = int(top[x, z])
        th = 4 + int(rng.integers(0, 3))
        t[x - 2:x + 3, y0 + th - 2:y0 + th, z - 2:z + 3] = B["leaves"]
        t[x - 1:x + 2, y0 + th:y0 + th + 2, z - 1:z + 2] = B["leaves"]
        t[x, y0:y0 + th, z] = B["log"]
    # hidden interior (hide_interior, on the slab: the cells above it are air, so its top layer is never hidden)
    opaque = np.zeros(256, bool)
    for k in ("bedrock", "stone", "dirt", "grass", "sand", "log", "plank", "glow", "snow", "flat"):
        opaque[B[k]] = True
    op = opaque[t]
    inner = op.copy()
    for ax in range(3):
        for sh in (1, -1):
            rr = np.roll(op, sh, axis=ax)
            sl = [slice(None)] * 3
            sl[ax] = 0 if sh == 1 else -1
            rr[tuple(sl)] = False
            inner &= rr
    t[inner] = ANY
    del op, inner
    octree = build_octree_slab(t, depth, ANY)
    cam = look_at_camera((n * 0.18, hi_h + 0.20 * height, n * 0.15), (n / 2, lo_h + 0.3 * (hi_h - lo_h), n * 0.55), fov_deg=70.0)
    return PackedScene(octree=octree, octree_depth=depth, block_palette=small.block_palette, material_palette=small.material_palette,
                       aabb_models=small.aabb_models, quad_models=small.quad_models, world_bvh=empty_bvh(), actor_bvh=empty_bvh(),
                       bvh_trigs=np.zeros(1, np.int32), atlas=small.atlas, sky=small.sky, sky_intensity=small.sky_intensity, sun=small.sun,
                       camera=cam, width=width, height=img_height, name=f"bigworld{chunks}x{chunks}")


def cached_big_outdoor_world(**kw) -> PackedScene:
    """big_outdoor_world(**kw) through the on-disk cache (a few minutes and ~12 GB of numpy to generate at 128 x 128 chunks)."""
    import hashlib
    import inspect
    import os
    bound = inspect.signature(big_outdoor_world).bind(**kw)
    bound.apply_defaults()
    src = inspect.getsource(big_outdoor_world) + inspect.getsource(build_octree_slab) + inspect.getsource(outdoor_world)
    key = hashlib.sha256((repr(sorted(bound.arguments.items())) + src).encode()).hexdigest()[:16]
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), ".scene_cache", f"bigworld_{key}.npz")
    if os.path.exists(path):
        try:
            return load_scene(path)
        except Exception:
            pass
    sc = big_outdoor_world(**kw)
    try:
        save_scene(sc, path)
    except OSError:
        pass
    return sc


def cached_entity_world(n_world_tris: int, actor_tris: int = 5000, seed: int = 11, region=((40, 90, 40), (470, 170, 470)), **world_kw) -> PackedScene:
    """BASELINE configs[4]: cached_outdoor_world(chunks=32, height=256, **world_kw) + add_entities(...) through the same on-disk
    cache (building the BVH of 10^6 triangles takes about a minute of numpy)."""
    import hashlib
    import inspect
    import os
    world_kw = dict(dict(chunks=32, height=256), **world_kw)
    src = inspect.getsource(add_entities) + inspect.getsource(outdoor_world) + inspect.getsource(build_octree) + inspect.getsource(hide_interior)
    key = hashlib.sha256((repr((n_world_tris, actor_tris, seed, region, sorted(world_kw.items()))) + src).encode()).hexdigest()[:16]
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), ".scene_cache")
    path = os.path.join(root, f"entities_{key}.npz")
    if os.path.exists(path):
        try:
            return load_scene(path)
        except Exception:
            pass
    sc = add_entities(cached_outdoor_world(**world_kw), n_world_tris, seed=seed, actor_tris=actor_tris, region=region)
    try:
        save_scene(sc, path)
    except OSError:
        pass
    return sc
