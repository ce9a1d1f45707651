"""The named small scenes behind tests/golden/*.npz (inputs are regenerated from seeds; outputs
were produced by the reference kernel itself, see tests/golden/generate.py)."""
import dataclasses
import hashlib

import numpy as np

from chunkyclplugin_amd import scenes

W, H = 64, 48
N_PASSES = 3


def _outdoor():
    return scenes.outdoor_world(chunks=2, height=48, seed=101, width=W, img_height=H, aabb_frac=0.08,
                                quad_frac=0.05, emitters=0.02)


def make(name: str) -> scenes.PackedScene:
    if name == "outdoor":
        return _outdoor()
    if name == "outdoor_nosun":
        return dataclasses.replace(_outdoor(), sun=scenes.pack_sun(0.6, 1.2, 1.25, False))
    if name == "entities":
        sc = _outdoor()
        return scenes.add_entities(sc, 600, seed=5, actor_tris=120, region=((2, 20, 2), (30, 44, 30)))
    if name == "dof":
        sc = _outdoor()
        cam = sc.camera.copy()
        cam[12], cam[13] = 0.08, 18.0
        return dataclasses.replace(sc, camera=cam)
    if name == "pregen":
        sc = _outdoor()
        rng = np.random.default_rng(9)
        rays = np.zeros((W * H, 6), np.float32)
        rays[:, :3] = sc.camera[:3]
        d = rng.normal(size=(W * H, 3))
        d[:, 1] -= 0.8
        rays[:, 3:] = d
        return dataclasses.replace(sc, camera=rays.reshape(-1), projector_type=-1)
    if name == "inside":
        sc = _outdoor()
        return dataclasses.replace(sc, camera=scenes.look_at_camera((10.3, 9.2, 12.1), (20, 14, 20), 90.0))
    if name == "atlas_layers":   # thirteen textures over four 32x32 atlas layers (quirk B#7: the 19-bit layer mask)
        return scenes.outdoor_world(chunks=2, height=48, seed=101, width=W, img_height=H, aabb_frac=0.08, quad_frac=0.05,
                                    emitters=0.02, atlas_tiles=(2, 2))
    if name == "water":          # tint type 3 (biome water, K/material.h:61-72) beside types 1, 2 and 0xFF
        return scenes.outdoor_world(chunks=2, height=48, seed=101, width=W, img_height=H, aabb_frac=0.08, quad_frac=0.05,
                                    emitters=0.02, water=True)
    if name == "indoor":
        return scenes.indoor_room(size=24, seed=7, width=W, img_height=H, emitter_frac=0.03)
    if name == "indoor_sun":
        sc = scenes.indoor_room(size=24, seed=7, width=W, img_height=H, emitter_frac=0.03)
        return dataclasses.replace(sc, sun=scenes.pack_sun(0.6, 1.2, 1.25, True))
    raise KeyError(name)


NAMES = ["outdoor", "outdoor_nosun", "entities", "dof", "pregen", "inside", "indoor", "indoor_sun", "atlas_layers", "water"]
RECORD_GIDS = np.arange(0, W * H, 37, dtype=np.int32)


def input_digest(sc: scenes.PackedScene) -> str:
    h = hashlib.sha256()
    for f in ("octree", "block_palette", "material_palette", "aabb_models", "quad_models", "world_bvh",
              "actor_bvh", "bvh_trigs", "atlas", "sky", "sun", "camera"):
        h.update(np.ascontiguousarray(getattr(sc, f)).tobytes())
    h.update(repr((sc.octree_depth, float(sc.sky_intensity), sc.projector_type, sc.width, sc.height)).encode())
    return h.hexdigest()


# ---- the BASELINE views at the sizes that are timed (tests/golden/timed_rows.npz: rows rendered by the reference build) ----
TIMED_PASSES = 8
TIMED_VIEWS = ["city", "city_entities", "outdoor", "indoor", "entities", "entities4k"]


def timed_view(name: str) -> scenes.PackedScene:
    """configs[1] (benchmark/OpenCL_test city, without and with its entities), configs[2] (32x32-chunk world: bench.py),
    configs[3] (indoor room), configs[4] (100 000 + 5 000 triangles) at 1920x1080 and at its stated 3840x2160 — exactly as
    bench.py / tools/config_bench.py / tests/test_gpu_timed_kernels.py build them."""
    from chunkyclplugin_amd import octree2
    if name == "city":
        return octree2.cached_benchmark_scene(1920, 1080)
    if name == "city_entities":
        return octree2.cached_benchmark_scene(1920, 1080, entities=True)
    if name == "outdoor":
        return scenes.cached_outdoor_world(chunks=32, height=256)
    if name == "indoor":
        return scenes.indoor_room(size=64, width=1920, img_height=1080)
    if name in ("entities", "entities4k"):
        sc = scenes.add_entities(scenes.cached_outdoor_world(chunks=32, height=256), 100000, seed=11, actor_tris=5000,
                                 region=((40, 90, 40), (470, 170, 470)))
        return sc.with_view(3840, 2160) if name == "entities4k" else sc
    raise KeyError(name)


def timed_rows(sc: scenes.PackedScene):
    """Sixteen whole rows per view, evenly spread from sky to foreground (30 720 pixels of a 1920-wide view, 61 440 at 3840)."""
    h = sc.height
    return [((2 * k + 1) * h) // 32 for k in range(16)]


# ---- helper-level known answers (tests/golden/helpers.npz: the reference object's own helpers on these rows) ----
HELPER_SCENE = "entities"
HELPER_KINDS = [0, 1, 2, 3, 4, 6, 7, 8, 9, 10, 11, 12, 14, 15]
HELPER_ROWS = 768


def _unit(v):
    return v / np.linalg.norm(v, axis=1, keepdims=True)


def _bits(ints):
    return np.asarray(ints, np.int32).view(np.float32)


def helper_rows(sc: scenes.PackedScene, which: int) -> np.ndarray:
    """Input rows (32 floats each; ints as bit patterns) for helper `which` — layouts in oracle/ref_shim.cpp ref_helpers."""
    n = HELPER_ROWS
    rng = np.random.default_rng(1000 + which)
    rows = np.zeros((n, 32), np.float32)
    f32 = np.float32
    if which in (0, 1, 2, 3):
        lo = rng.uniform(0, 0.6, (n, 3))
        hi = lo + rng.uniform(0.05, 0.4, (n, 3))
        rows[:, 0:6:2], rows[:, 1:6:2] = lo, hi
        o = rng.uniform(-0.5, 1.5, (n, 3))
        target = np.where(rng.random((n, 1)) < 0.7, lo + rng.uniform(0, 1, (n, 3)) * (hi - lo), rng.uniform(0, 1, (n, 3)))
        d = _unit(target - o)
        d[::7, 0] = 0.0                       # axis-parallel rays: invDir = +-inf, NaN products at the slab planes
        d[3::11, 1] = -0.0
        o[5::13] = lo[5::13]                   # origins exactly on a corner / a face
        o[6::17, 2] = hi[6::17, 2]
        if which == 2:                         # unit box: the caller passes the MARCH POSITION as `dir` (K/block.h:52)
            rows[:, 9:12] = rng.uniform(0, 64, (n, 3))
            rows[:, 12:15] = d
            o = rng.uniform(-0.2, 1.2, (n, 3))
            o[::5] = np.where(rng.random((len(o[::5]), 3)) < 0.5, 0.0, 1.0) + rng.uniform(-1e-4, 1e-4, (len(o[::5]), 3))
        else:
            rows[:, 9:12] = d
        rows[:, 6:9] = o
    elif which == 4:
        blocks = np.asarray(sc.block_palette, np.int32).reshape(-1, 2)
        kinds = [np.flatnonzero(blocks[:, 0] == t) for t in (1, 2, 3)]
        pick = np.concatenate([rng.choice(k, n // 3 + 1) for k in kinds if len(k)])[:n]
        rng.shuffle(pick)
        rows[:, 0] = _bits(2 * pick)          # block pointer = 2 * palette index
        cell = rng.integers(0, 30, (n, 3))
        rows[:, 1:4] = cell
        target = cell + rng.uniform(0.05, 0.95, (n, 3))
        face = rng.integers(0, 3, n)
        pos = cell + rng.uniform(-0.05, 1.05, (n, 3))
        side = rng.integers(0, 2, n)
        pos[np.arange(n), face] = cell[np.arange(n), face] + side + np.where(side == 1, 1e-4, -1e-4)  # just outside a face
        pos[::9] = cell[::9] + rng.uniform(0.2, 0.8, (len(pos[::9]), 3))                              # inside the cell
        rows[:, 4:7] = pos
        rows[:, 7:10] = _unit(target - pos)
    elif which == 6:
        o = rng.uniform(0, 8, (n, 3))
        e1, e2 = rng.normal(size=(n, 3)), rng.normal(size=(n, 3))
        flags = rng.integers(0, 2, n) << 8
        rows[:, 0] = _bits(flags)
        rows[:, 1:4], rows[:, 4:7], rows[:, 7:10] = e1, e2, o
        rows[:, 10:13] = _unit(np.cross(e1, e2))
        rows[:, 13:19] = rng.uniform(0, 1, (n, 6))
        rows[:, 19] = _bits(6 * rng.integers(0, 5, n))
        bary = rng.uniform(0, 1, (n, 2))
        fold = bary.sum(axis=1) > 1
        bary[fold] = 1 - bary[fold]            # inside the triangle
        bary[::4] = rng.uniform(-0.3, 1.3, (len(bary[::4]), 2))   # and around its edges
        hitp = o + e1 * bary[:, :1] + e2 * bary[:, 1:]
        ro = hitp + rng.normal(size=(n, 3)) * 3
        rows[:, 20:23] = ro
        rows[:, 23:26] = _unit(hitp - ro)
        rows[:, 26] = np.where(rng.random(n) < 0.8, np.inf, rng.uniform(0.5, 4, n))
    elif which in (7, 10):
        rows[:, 0] = _bits(rng.integers(-2**31, 2**31 - 1, n))
        nrm = _unit(rng.normal(size=(n, 3)))
        axes = np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]], f32)
        nrm[::3] = axes[rng.integers(0, 6, len(nrm[::3]))]
        nrm[1::16, 0] = 0.1                    # the double compare |n.x| > 0.1 of K/kernel.h:66 at its threshold
        rows[:, 1:4] = nrm
        rows[:, 4:7] = rng.uniform(0, 32, (n, 3))
    elif which in (8, 9):
        d = _unit(rng.normal(size=(n, 3)))
        if which == 8:                         # around the sun: inside and outside the textured disc
            sun = np.asarray(sc.sun, np.int32).view(np.float32)
            alt, az = float(sun[4]), float(sun[5])
            sw = np.array([np.cos(az) * abs(np.cos(alt)), np.sin(alt), np.sin(az) * abs(np.cos(alt))])
            d[: n * 3 // 4] = _unit(sw + rng.normal(size=(n * 3 // 4, 3)) * 0.08)
            rows[:, 3:7] = rng.uniform(0, 1, (n, 4))
            rows[::4, 3:7] = np.array([-0.0, 0.0, -0.0, 0.0], f32)
        else:
            axes = np.array([[1, 0, 0], [0, 1, 0], [0, 0, 1], [-1, 0, 0], [0, -1, 0], [0, 0, -1]], f32)
            d[::8] = axes[np.arange(len(d[::8])) % 6]   # atan2 / asin at their special arguments
        rows[:, 0:3] = d
    elif which in (11, 12):
        mats = np.asarray(sc.material_palette, np.int32).reshape(-1, 6)
        tex = np.flatnonzero(mats[:, 0] & 4)
        if which == 11:
            m = mats[rng.choice(tex, n)]
            rows[:, 0:2] = rng.uniform(-0.1, 1.1, (n, 2))
            rows[::6, 0], rows[1::6, 1] = 1.0, 0.0
            rows[:, 2], rows[:, 3] = _bits(m[:, 3]), _bits(m[:, 2])
        else:
            rows[:, 0] = _bits(6 * rng.integers(0, len(mats), n))
            rows[:, 1:3] = rng.uniform(0, 1, (n, 2))
    elif which in (14, 15):
        cam = np.asarray(sc.camera, np.float32)
        side = 1 << int(sc.octree_depth)
        o = np.tile(cam[:3], (n, 1)).astype(np.float64)
        d = _unit(rng.uniform((0, 0, 0), (side * 0.5, side * 0.35, side * 0.5), (n, 3)) - o)   # towards the terrain
        o[n // 3:] = rng.uniform(0, side * 0.5, (n - n // 3, 3)) * np.array([1, 1.4, 1])   # bounce rays from inside the world
        o[::10] = rng.uniform(-20, side + 20, (len(o[::10]), 3))                       # and from outside it
        d[n // 3:] = _unit(rng.normal(size=(n - n // 3, 3)) + np.array([0.0, -0.5, 0.0]))
        d[::12, 1] = 0.0
        rows[:, 0:3], rows[:, 3:6] = o, d
        rows[:, 6] = np.where(rng.random(n) < 0.7, np.inf, rng.uniform(1, 30, n))
        if which == 15:                        # aim most rays at the entities' region
            rows[: n * 2 // 3, 3:6] = _unit(rng.uniform((2, 20, 2), (30, 44, 30), (n * 2 // 3, 3)) - rows[: n * 2 // 3, 0:3])
    else:
        raise KeyError(which)
    return rows


def rows_digest(rows: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(rows, np.float32).tobytes()).hexdigest()


# ---- the +z face of an AABB model (K/primitives.h:209-234 leaves its material unset; the reference build takes EAST) ----
PLUS_Z_EAST = np.array([0, 255 / 256, 0, 255 / 256], np.float32)   # colour of the east material below, as colorFromArgb gives it
PLUS_Z_GID = 4 * 9 + 4                                              # the centre pixel looks at the +z face


def plus_z_scene() -> scenes.PackedScene:
    """One AABB-model block with six differently coloured face materials, seen along -z at its +z face."""
    pal = scenes.Palettes()
    cols = [0xFFFF0000, 0xFF00FF00, 0xFF0000FF, 0xFFFFFF00, 0xFFFF00FF, 0xFF00FFFF]
    mats = [pal.material(argb=c) for c in cols]
    pal.block_invisible()
    blk = pal.block_aabbs([((0.25, 0.75, 0.25, 0.75, 0.25, 0.75), 0, tuple(mats))])
    t = np.zeros((8, 8, 8), np.int32)
    t[4, 4, 4] = blk
    b, m, a, q = pal.arrays()
    return scenes.PackedScene(octree=scenes.build_octree(t, 3), octree_depth=3, block_palette=b, material_palette=m,
                            aabb_models=a, quad_models=q, world_bvh=scenes.empty_bvh(), actor_bvh=scenes.empty_bvh(),
                            bvh_trigs=np.zeros(1, np.int32), atlas=np.zeros((1, 16, 16, 4), np.uint8),
                            sky=scenes.bake_sky(16), sky_intensity=1.0, sun=scenes.pack_sun(0.6, 1.2, 1.0, False),
                            camera=scenes.look_at_camera((4.5, 4.5, 7.5), (4.5, 4.5, 4.5), 40.0), width=9, height=9)


# ---- the scene of tools/cull_probe.py / tests/test_bvh_cull.py: entity boxes ON the block grid ----
GRID_BOXES = 1500


def grid_boxes(seed=5):
    """tiny_scene's palettes with a world BVH of GRID_BOXES axis-aligned boxes with integer corners (12 triangles each, a third of
    the boxes two-sided): every face lies in a plane that rays leave from.  Returns (scene, box minima, box maxima)."""
    import dataclasses
    base = scenes.tiny_scene(seed=3, size=16, width=16, height=8, entities=0)
    rng = np.random.default_rng(seed)
    nmat = len(base.material_palette) // 6
    lo = rng.integers(2, 60, size=(GRID_BOXES, 3)).astype(np.float64)
    hi = lo + rng.integers(1, 4, size=(GRID_BOXES, 3))
    tris = []
    for b in range(GRID_BOXES):
        x0, y0, z0 = lo[b]
        x1, y1, z1 = hi[b]
        v = np.array([[x0, y0, z0], [x1, y0, z0], [x1, y1, z0], [x0, y1, z0], [x0, y0, z1], [x1, y0, z1], [x1, y1, z1], [x0, y1, z1]])
        mat = 6 * int(rng.integers(0, nmat))
        ds = bool(rng.random() < 0.33)
        for (i, j, k) in ((0, 2, 1), (0, 3, 2), (4, 5, 6), (4, 6, 7), (0, 1, 5), (0, 5, 4), (3, 6, 2), (3, 7, 6), (0, 4, 7), (0, 7, 3), (1, 2, 6), (1, 6, 5)):
            tris.append(scenes.pack_triangle(v[i], v[j], v[k], (0, 0), (1, 0), (0, 1), mat, ds))
    t = np.array(tris, np.int64).astype(np.int32).reshape(-1, 20)
    wn, wtr = scenes.build_bvh(t, 4)
    sc = dataclasses.replace(base, world_bvh=wn, actor_bvh=scenes.empty_bvh(), bvh_trigs=wtr, name="grid_boxes")
    return sc, lo.astype(np.float32), hi.astype(np.float32)
