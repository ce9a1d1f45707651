"""Reader for Chunky scene dumps (`<scene>.octree2` + `<scene>.json`) — SURVEY.md section 8 row f1.

The reference gets its octree from Chunky's `PackedOctree.treeData` in memory
(`AbstractSceneLoader.java:148-159`, `ClSceneLoader.java:52-63`).  Without Chunky, the same tree can
be read from the `.octree2` file Chunky writes next to a scene (format reconstructed in SURVEY.md
Appendix D): gzip, big-endian `DataOutputStream`:

    int version (6) | int paletteVersion (4) | int nBlocks | nBlocks x NBT compound payload
    world octree:  int depth | pre-order node stream (-1 = branch followed by its 8 children in
                   (x<<2 | y<<1 | z) order, otherwise leaf = palette index, 0x7FFFFFFE = ANY_TYPE)
    water octree:  same (skipped) | biome textures (skipped)

Block models and textures come from Chunky + Minecraft assets, which are not available, so they are
procedural (`asset_pack`): a palette entry gets the model class its NAME and PROPERTIES say it has — slab
(bottom / top / double), stairs (half, facing), fence / wall / bars / pane (post + connected arms), carpet /
rail / plate / trapdoor, door / gate / ladder, crossed-quad plants, small boxes for torches and the like,
a full cube otherwise — and each block name one 16 x 16 texture hashed from the name (BASELINE.md section 4,
config 1/2: "16x16 hashed textures"); leaves, plants, bars and panes have cut-out texels.
"""
from __future__ import annotations

import gzip
import json
import os
import math
import struct
import zlib
from typing import List, Tuple

import numpy as np

from . import scenes

ANY_TYPE = 0x7FFFFFFE


class _Reader:
    def __init__(self, data: bytes):
        self.d, self.p = data, 0

    def i8(self):
        v = self.d[self.p]
        self.p += 1
        return v

    def u16(self):
        v = struct.unpack_from(">H", self.d, self.p)[0]
        self.p += 2
        return v

    def i32(self):
        v = struct.unpack_from(">i", self.d, self.p)[0]
        self.p += 4
        return v

    def skip(self, n):
        self.p += n

    def string(self):
        n = self.u16()
        s = self.d[self.p:self.p + n].decode("utf-8", "replace")
        self.p += n
        return s

    def payload(self, tag):
        """NBT payload of type `tag`; returns a Python value (compounds as dicts)."""
        if tag == 1:
            return self.i8()
        if tag == 2:
            return self.u16()
        if tag == 3:
            return self.i32()
        if tag == 4:
            self.skip(8)
            return None
        if tag == 5:
            self.skip(4)
            return None
        if tag == 6:
            self.skip(8)
            return None
        if tag == 7:
            self.skip(self.i32())
            return None
        if tag == 8:
            return self.string()
        if tag == 9:
            t, n = self.i8(), self.i32()
            return [self.payload(t) for _ in range(n)]
        if tag == 10:
            return self.compound()
        if tag == 11:
            self.skip(4 * self.i32())
            return None
        if tag == 12:
            self.skip(8 * self.i32())
            return None
        raise ValueError(f"unknown NBT tag {tag} at {self.p}")

    def compound(self):
        out = {}
        while True:
            tag = self.i8()
            if tag == 0:
                return out
            name = self.string()
            out[name] = self.payload(tag)


def read_octree2(path: str) -> Tuple[List[dict], int, np.ndarray]:
    """Returns (block palette as NBT dicts, depth, pre-order node stream as int32)."""
    r = _Reader(gzip.open(path).read())
    version = r.i32()
    if version not in (5, 6):
        raise ValueError(f"unsupported .octree2 version {version}")
    r.i32()  # palette version
    n_blocks = r.i32()
    palette = [r.compound() for _ in range(n_blocks)]
    depth = r.i32()
    # the pre-order stream ends when the root subtree is complete
    stream = np.frombuffer(r.d, dtype=">i4", offset=r.p)
    need, i = 1, 0
    # walk in chunks with numpy: every -1 adds 8 pending nodes, every node consumes one
    n = stream.size
    pos = 0
    CH = 1 << 20
    while need > 0:
        blk = stream[pos:pos + CH]
        if blk.size == 0:
            raise ValueError("truncated octree stream")
        delta = np.where(blk == -1, 7, -1).astype(np.int64)
        run = need + np.cumsum(delta)
        done = np.nonzero(run == 0)[0]
        if done.size:
            pos += int(done[0]) + 1
            need = 0
        else:
            need = int(run[-1])
            pos += blk.size
    return palette, depth, stream[:pos].astype(np.int32)


def pack_preorder(stream: np.ndarray, n_types: int) -> np.ndarray:
    """Pre-order node stream -> the reference's `octreeData`: index 0 = root,
    v > 0 = index of an 8-int child group, v <= 0 = -(2 * palette index) (block pointer, each block
    packs to 2 ints), ANY_TYPE kept.  Groups are allocated depth-first like Chunky's loader."""
    n_branch = int((stream == -1).sum())
    data = np.zeros(1 + 8 * n_branch, np.int64)
    nxt = 1
    # explicit stack of (slot to fill) in reverse child order
    stack = [0]
    s = stream.tolist()
    for v in s:
        slot = stack.pop()
        if v == -1:
            data[slot] = nxt
            stack.extend(range(nxt + 7, nxt - 1, -1))
            nxt += 8
        elif v == ANY_TYPE or v >= n_types:
            data[slot] = -ANY_TYPE
        else:
            data[slot] = -2 * v
    assert not stack and nxt == data.size
    return data.astype(np.int32)


def _block_color(name: str) -> int:
    h = zlib.crc32(name.encode())
    r, g, b = 64 + (h & 0x7F), 64 + ((h >> 8) & 0x7F), 64 + ((h >> 16) & 0x7F)
    # a few recognisable overrides so the city reads as a city
    table = {"stone": (125, 125, 125), "grass_block": (110, 160, 80), "dirt": (134, 96, 67), "water": (60, 100, 200),
             "sand": (219, 207, 163), "oak_leaves": (70, 130, 50), "glass": (200, 220, 230), "bedrock": (60, 60, 60),
             "white_concrete": (220, 220, 220), "gray_concrete": (90, 90, 90), "black_concrete": (30, 30, 35)}
    key = name.split(":")[-1]
    if key in table:
        r, g, b = table[key]
    return 0xFF000000 | (r << 16) | (g << 8) | b


INVISIBLE = {"minecraft:air", "minecraft:cave_air", "minecraft:void_air", "minecraft:barrier", "minecraft:structure_void"}

# ---------------------------------------------------------------------------------------- procedural asset pack
# Chunky packs every palette entry through its block model and its textures (AbstractSceneLoader.java:101-127, PackedBlock /
# PackedAabb / PackedQuad); neither chunky-core's models nor Minecraft's textures exist in this image.  What does exist is the
# palette itself: 3 091 entries with their NAMES and PROPERTIES (slab type, stair half / facing, fence connections ...).  The
# pack below gives each entry the model class its name and properties say it has — with Minecraft's well-known block geometry
# for the common classes — and each block NAME one 16 x 16 texture hashed from the name (leaves and plants with cut-out texels).
# It is a stand-in with the right KIND of geometry and the right mix of model blocks, not Chunky's own output.
_PLANTS = ("grass", "tall_grass", "fern", "large_fern", "dead_bush", "dandelion", "poppy", "blue_orchid", "allium", "azure_bluet",
           "oxeye_daisy", "cornflower", "lily_of_the_valley", "wither_rose", "sunflower", "lilac", "rose_bush", "peony", "sugar_cane",
           "wheat", "carrots", "potatoes", "beetroots", "nether_wart", "sweet_berry_bush", "cobweb", "brown_mushroom", "red_mushroom",
           "seagrass", "tall_seagrass", "kelp", "kelp_plant", "bamboo", "crimson_roots", "warped_roots", "nether_sprouts",
           "crimson_fungus", "warped_fungus", "twisting_vines", "weeping_vines", "twisting_vines_plant", "weeping_vines_plant")
_FLAT = ("rail", "powered_rail", "detector_rail", "activator_rail", "lily_pad", "redstone_wire", "repeater", "comparator", "snow")


def _model_class(key: str, props: dict):
    """(class, parameters) of a palette entry: cube | slab | stairs | thin | post | door | plant | small."""
    if key.endswith("_slab"):
        return ("slab", props.get("type", "bottom"))
    if key.endswith("_stairs"):
        return ("stairs", props.get("half", "bottom"), props.get("facing", "north"))
    if key.endswith("_carpet") or key.endswith("_pressure_plate") or key in _FLAT:
        return ("thin", 0.0625, "bottom")
    if key.endswith("_trapdoor"):
        if props.get("open", "false") == "true":
            return ("door", props.get("facing", "north"), 0.1875)
        return ("thin", 0.1875, props.get("half", "bottom"))
    if key.endswith("_door") or key.endswith("_fence_gate") or key == "ladder" or key.endswith("_wall_sign") or key == "vine":
        return ("door", props.get("facing", "north"), 0.1875 if key.endswith("_door") else 0.125)
    if key.endswith("_fence") or key.endswith("_wall") or key in ("iron_bars", "chain", "end_rod", "lightning_rod") or key.endswith("glass_pane"):
        half = 0.25 if key.endswith("_wall") else (0.125 if key.endswith("_fence") else 0.0625)
        sides = tuple(d for d in ("north", "east", "south", "west") if props.get(d, "false") not in ("false", "none"))
        return ("post", half, sides, key.endswith("_fence"))
    if key in _PLANTS or key.endswith("_sapling") or key.endswith("_tulip") or key.endswith("_coral") or key.endswith("_coral_fan"):
        return ("plant",)
    if key.endswith("torch") or key.endswith("lantern") or key.endswith("_candle") or key == "candle" or key.endswith("_button") or \
            key == "lever" or key.endswith("_head") or key.endswith("_skull") or key == "flower_pot" or key.startswith("potted_"):
        return ("small",)
    return ("cube",)


def _model_boxes(cls):
    """AABB boxes (xmin, xmax, ymin, ymax, zmin, zmax) of a model class — Minecraft's familiar shapes."""
    kind = cls[0]
    if kind == "slab":
        return {"bottom": [(0, 1, 0, 0.5, 0, 1)], "top": [(0, 1, 0.5, 1, 0, 1)]}.get(cls[1])   # "double" -> None: a full cube
    if kind == "stairs":
        lo, hi = ((0, 0.5), (0.5, 1)) if cls[1] == "bottom" else ((0.5, 1), (0, 0.5))
        step = {"east": (0.5, 1, hi[0], hi[1], 0, 1), "west": (0, 0.5, hi[0], hi[1], 0, 1),
                "south": (0, 1, hi[0], hi[1], 0.5, 1), "north": (0, 1, hi[0], hi[1], 0, 0.5)}[cls[2] if cls[2] in ("east", "west", "south", "north") else "north"]
        return [(0, 1, lo[0], lo[1], 0, 1), step]
    if kind == "thin":
        t = cls[1]
        return [(0, 1, 1 - t, 1, 0, 1)] if cls[2] == "top" else [(0, 1, 0, t, 0, 1)]
    if kind == "door":
        t = cls[2]
        return [{"north": (0, 1, 0, 1, 1 - t, 1), "south": (0, 1, 0, 1, 0, t), "west": (1 - t, 1, 0, 1, 0, 1), "east": (0, t, 0, 1, 0, 1)}
                .get(cls[1], (0, 1, 0, 1, 0, t))]
    if kind == "post":
        h, sides, fence = cls[1], cls[2], cls[3]
        boxes = [(0.5 - h, 0.5 + h, 0, 1, 0.5 - h, 0.5 + h)]
        y0, y1, w = (0.375, 0.9375, 0.0625) if fence else (0.0, 0.875 if h == 0.25 else 1.0, 0.1875 if h == 0.25 else 0.0625)
        arm = {"north": (0.5 - w, 0.5 + w, y0, y1, 0, 0.5 - h), "south": (0.5 - w, 0.5 + w, y0, y1, 0.5 + h, 1),
               "west": (0, 0.5 - h, y0, y1, 0.5 - w, 0.5 + w), "east": (0.5 + h, 1, y0, y1, 0.5 - w, 0.5 + w)}
        return boxes + [arm[d] for d in sides]
    if kind == "small":
        return [(0.375, 0.625, 0, 0.625, 0.375, 0.625)]
    return None


def asset_pack(palette: List[dict], emitters: bool = False):
    """(Palettes, atlas, texture records incl. the sun's at [-1]) for an `.octree2` block palette."""
    names = []
    for blk in palette:
        n = blk.get("Name", "minecraft:air")
        if n not in names:
            names.append(n)
    tiles = int(math.ceil(math.sqrt(len(names) + 8)))
    ab = scenes.AtlasBuilder(tiles, tiles)
    tex = {}
    for n in names:
        key = n.split(":")[-1]
        argb = _block_color(n)
        rgb = ((argb >> 16) & 255, (argb >> 8) & 255, argb & 255)
        cls = _model_class(key, {})
        holes = 0.45 if cls[0] == "plant" else (0.2 if key.endswith("_leaves") or key in ("cobweb", "iron_bars") or key.endswith("glass_pane") else 0.0)
        tex[n] = ab.add(scenes.noise_texture(np.random.default_rng(zlib.crc32(n.encode())), rgb, 18, holes=holes))
    tsun = ab.add(scenes.noise_texture(np.random.default_rng(1), (255, 250, 230), 4, size=32))
    atlas, recs = ab.build()
    pal = scenes.Palettes()
    mat = {}
    models = {}   # model class -> pointer into the AABB / quad palette (entries of one class and material share a model)
    for blk in palette:
        n = blk.get("Name", "minecraft:air")
        if n in INVISIBLE:
            pal.block_invisible()
            continue
        key = n.split(":")[-1]
        if n not in mat:
            glow = any(k in key for k in ("lantern", "glowstone", "torch", "lamp", "lava", "sea_pickle", "shroomlight", "magma"))
            tint = (1 << 24) if key.endswith("_leaves") or key == "vine" else ((2 << 24) if key in ("grass", "tall_grass", "fern", "large_fern") else 0)
            mat[n] = pal.material(texture=recs[tex[n]], tint=tint, emittance=1.0 if emitters and glow else 0.0)
        m = mat[n]
        cls = _model_class(key, blk.get("Properties", {}) or {})
        if cls[0] == "plant":
            if (cls, m) not in models:
                def quad(o, xv, yv):
                    return (o, xv, yv, (0.0, 1.0, 0.0, 1.0), m, 1)
                pal.block_quads([quad((0.15, 0, 0.15), (0.7, 0, 0.7), (0, 1, 0)), quad((0.85, 0, 0.85), (-0.7, 0, -0.7), (0, 1, 0)),
                                 quad((0.15, 0, 0.85), (0.7, 0, -0.7), (0, 1, 0)), quad((0.85, 0, 0.15), (-0.7, 0, 0.7), (0, 1, 0))])
                models[(cls, m)] = tuple(pal.blocks[-2:])
            else:
                pal.blocks += list(models[(cls, m)])
            continue
        boxes = _model_boxes(cls)
        if boxes is None:
            pal.block_cube(m)
        elif (cls, m) not in models:
            pal.block_aabbs([(b, 0, (m,) * 6) for b in boxes])
            models[(cls, m)] = tuple(pal.blocks[-2:])
        else:
            pal.blocks += list(models[(cls, m)])
    return pal, atlas, recs, tsun


def camera_from_json(cam: dict, origin=(0.0, 0.0, 0.0)) -> np.ndarray:
    """15 camera floats (`ClCamera.java:42-52`) from Chunky's scene JSON camera block: position minus
    the octree origin, rotation from yaw/pitch/roll, fovTan = 2 tan(fov/2)."""
    pos = cam["position"]
    o = cam["orientation"]
    yaw, pitch, roll = o["yaw"], o["pitch"], o["roll"]

    def rx(a):
        c, s = math.cos(a), math.sin(a)
        return np.array([[1, 0, 0], [0, c, -s], [0, s, c]])

    def ry(a):
        c, s = math.cos(a), math.sin(a)
        return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])

    def rz(a):
        c, s = math.cos(a), math.sin(a)
        return np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])

    # Chunky's Camera.updateTransform composes rotY(pi/2 + yaw) * rotX(pi/2 - pitch) * rotZ(roll) (pitch -pi/2 =
    # level); chunky-core is not available to confirm, this is the combination that renders the benchmark
    # scene upright from its saved camera
    M = ry(math.pi / 2 + yaw) @ rx(math.pi / 2 - pitch) @ rz(roll)
    p = np.array([pos["x"] - origin[0], pos["y"] - origin[1], pos["z"] - origin[2]])
    fov_tan = 2.0 * math.tan(math.radians(min(max(cam.get("fov", 70.0), 0.0), 180.0)) / 2.0)
    dof = cam.get("dof", "Infinity")
    subject = cam.get("focalOffset", 2.0)
    aperture = 0.0 if dof in ("Infinity", float("inf")) else subject / float(dof)
    return np.concatenate([p, M.reshape(-1), [aperture, subject, fov_tan]]).astype(np.float32)


def load_scene(octree2_path: str, json_path: str = None, width: int = 1920, height: int = 1080,
               emitters: bool = False) -> scenes.PackedScene:
    """`.octree2` (+ scene JSON for camera / sun) -> PackedScene through the procedural asset pack (`asset_pack`: model classes
    from names and properties, one hashed 16 x 16 texture per block name); `flat=True` in the environment of old fixtures is gone."""
    palette, depth, stream = read_octree2(octree2_path)
    pal, atlas, recs, tsun = asset_pack(palette, emitters)
    tree = pack_preorder(stream, len(palette))
    blocks, mats, aabbs, quads = pal.arrays()
    alt, azi, inten, draw = 0.6, 1.2, 1.25, True
    S = float(1 << depth)
    cam = scenes.look_at_camera((0.18 * S, 0.14 * S, 0.09 * S), (0.3 * S, 0.06 * S, 0.25 * S), 70.0)
    if json_path:
        js = json.loads(open(json_path, encoding="latin-1").read())
        sun = js.get("sun", {})
        alt, azi = sun.get("altitude", alt), sun.get("azimuth", azi)
        inten, draw = sun.get("intensity", inten), sun.get("drawTexture", draw)
        # Chunky subtracts the octree origin from the camera (ClCamera.java:39-40): origin = min chunk * 16, yMin
        chunks = js.get("chunkList", [])
        if chunks:
            ox = 16 * min(c[0] for c in chunks)
            oz = 16 * min(c[1] for c in chunks)
            origin = (float(ox), float(js.get("yMin", 0)), float(oz))
            cam = camera_from_json(js["camera"], origin)
    return scenes.PackedScene(octree=tree, octree_depth=depth, block_palette=blocks, material_palette=mats,
                              aabb_models=aabbs, quad_models=quads, world_bvh=scenes.empty_bvh(),
                              actor_bvh=scenes.empty_bvh(), bvh_trigs=np.zeros(1, np.int32), atlas=atlas,
                              sky=scenes.bake_sky(128, scenes.sun_direction(alt, azi)), sky_intensity=float(inten),
                              sun=scenes.pack_sun(alt, azi, inten, bool(draw), recs[tsun]), camera=cam, width=width,
                              height=height, name="octree2:" + octree2_path.split("/")[-1])


# ---------------------------------------------------------------------------------------- entities
# The scene JSON lists 4 188 entities (paintings, wall signs, banners, heads, standing signs) and 389 actors (armour
# stands) by kind, position and orientation.  Their models and textures are Chunky's + Minecraft's (not available), so
# each becomes a PROXY: one or a few flat-coloured boxes of about the real size at the real place, packed as triangles
# (PackedTriangle.java:72-78) into a world BVH and an actor BVH (PackedBvhNode.java:16-31).  That is enough to exercise
# the entity path of K/bvh.h on the benchmark scene's own entity distribution; it is not Chunky's geometry.
ENTITY_KINDS = ("painting", "wallsign", "wall_banner", "head", "standing_banner", "sign", "skull", "armor_stand")
_ENTITY_COLORS = (0xFFB08050, 0xFF9C7A4A, 0xFFC03030, 0xFFD0B090, 0xFF3050C0, 0xFF9C7A4A, 0xFFE0E0E0, 0xFF8A6A3A)


def entity_table(js: dict, origin) -> np.ndarray:
    """[n, 6] float32 rows {kind index, x, y, z (octree coordinates), yaw in radians, actor flag} from the scene JSON."""
    rows = []
    for actor, lst in ((0, js.get("entities", [])), (1, js.get("actors", []))):
        for e in lst:
            kind = e.get("kind")
            if kind not in ENTITY_KINDS:
                continue
            p = e.get("position", {})
            x, y, z = (float(p.get(k, 0.0)) - float(o) for k, o in zip("xyz", origin))
            if "angle" in e:
                yaw = math.radians(float(e["angle"]))
            elif "direction" in e and kind == "wallsign":           # 2 north, 3 south, 4 west, 5 east
                yaw = {2: math.pi, 3: 0.0, 4: math.pi / 2, 5: -math.pi / 2}.get(int(e["direction"]), 0.0)
            elif "direction" in e:
                yaw = float(e["direction"]) * math.pi / 8
            elif "rotation" in e:
                yaw = float(e["rotation"]) * math.pi / 8
            elif "pose" in e:
                allp = e["pose"].get("all") or [0.0, 0.0, 0.0]
                yaw = float(allp[1]) if len(allp) > 1 else 0.0
            else:
                yaw = 0.0
            rows.append((ENTITY_KINDS.index(kind), x, y, z, yaw, actor))
    return np.array(rows, np.float32).reshape(-1, 6)


def _box_tris(center, half, yaw, material):
    """12 single-sided triangles (outward faces) of a box rotated by `yaw` about the vertical through its centre."""
    c, s = math.cos(yaw), math.sin(yaw)
    corners = []
    for dx in (-1, 1):
        for dy in (-1, 1):
            for dz in (-1, 1):
                lx, ly, lz = dx * half[0], dy * half[1], dz * half[2]
                corners.append((center[0] + c * lx + s * lz, center[1] + ly, center[2] - s * lx + c * lz))
    v = lambda i, j, k: corners[(i << 2) | (j << 1) | k]
    quads = [((0, 0, 0), (0, 0, 1), (0, 1, 1), (0, 1, 0)), ((1, 0, 0), (1, 1, 0), (1, 1, 1), (1, 0, 1)),   # -x, +x
             ((0, 0, 0), (1, 0, 0), (1, 0, 1), (0, 0, 1)), ((0, 1, 0), (0, 1, 1), (1, 1, 1), (1, 1, 0)),   # -y, +y
             ((0, 0, 0), (0, 1, 0), (1, 1, 0), (1, 0, 0)), ((0, 0, 1), (1, 0, 1), (1, 1, 1), (0, 1, 1))]   # -z, +z
    out = []
    for q in quads:
        p0, p1, p2, p3 = (v(*k) for k in q)
        out.append(scenes.pack_triangle(p0, p1, p2, (0, 0), (1, 0), (1, 1), material))
        out.append(scenes.pack_triangle(p0, p2, p3, (0, 0), (1, 1), (0, 1), material))
    return out


def entity_proxies(table: np.ndarray, materials: List[int]):
    """(world triangles, actor triangles) as int32 [n, 20]; `materials` = one material pointer per ENTITY_KINDS entry."""
    world, actors = [], []
    for kind, x, y, z, yaw, actor in table:
        k, m = int(kind), materials[int(kind)]
        name = ENTITY_KINDS[k]
        dst = actors if actor else world
        if name == "painting":
            dst += _box_tris((x, y, z), (0.5, 0.5, 0.03), yaw, m)
        elif name == "wallsign":
            dst += _box_tris((x + 0.5, y + 0.55, z + 0.5), (0.5, 0.25, 0.045), yaw, m)
        elif name == "wall_banner":
            dst += _box_tris((x + 0.5, y - 0.1, z + 0.5), (0.45, 0.9, 0.04), yaw, m)
        elif name in ("head", "skull"):
            dst += _box_tris((x + 0.5, y + 0.25, z + 0.5), (0.25, 0.25, 0.25), yaw, m)
        elif name == "standing_banner":
            dst += _box_tris((x + 0.5, y + 0.95, z + 0.5), (0.04, 0.95, 0.04), yaw, m)
            dst += _box_tris((x + 0.5, y + 1.0, z + 0.5), (0.45, 0.8, 0.03), yaw, m)
        elif name == "sign":
            dst += _box_tris((x + 0.5, y + 0.3, z + 0.5), (0.045, 0.3, 0.045), yaw, m)
            dst += _box_tris((x + 0.5, y + 0.85, z + 0.5), (0.5, 0.25, 0.045), yaw, m)
        elif name == "armor_stand":
            dst += _box_tris((x, y + 0.03, z), (0.375, 0.03, 0.375), yaw, m)          # base plate
            for side in (-0.12, 0.12):                                                 # legs
                dst += _box_tris((x + side * math.cos(yaw), y + 0.4, z - side * math.sin(yaw)), (0.06, 0.37, 0.06), yaw, m)
            dst += _box_tris((x, y + 1.05, z), (0.25, 0.28, 0.09), yaw, m)            # body
            dst += _box_tris((x, y + 1.6, z), (0.2, 0.2, 0.2), yaw, m)                # head
    arr = lambda lst: np.array(lst, np.int64).astype(np.int32).reshape(-1, 20)
    return arr(world), arr(actors)


def with_entities(sc: scenes.PackedScene, table: np.ndarray) -> scenes.PackedScene:
    """The scene with the proxies of `table` in its world / actor BVHs (flat-colour materials appended to the palette)."""
    mats = list(np.asarray(sc.material_palette).tolist())
    ptrs = []
    for argb in _ENTITY_COLORS:
        ptrs.append(len(mats))
        mats += [0, 0, 0, argb - (1 << 32) if argb >= (1 << 31) else argb, 0, 0]
    world, actors = entity_proxies(table, ptrs)
    wn, wtr = scenes.build_bvh(world, 4)
    if len(actors):
        an, atr = scenes.build_bvh(actors, 4)
        an = an.copy().reshape(-1, 7)
        leaf = an[:, 0] <= 0
        an[leaf, 0] -= len(wtr)                        # one shared triangle array, as AbstractSceneLoader.java:118-127
        an, trigs = an.reshape(-1), np.concatenate([wtr, atr])
    else:
        an, trigs = scenes.empty_bvh(), wtr
    import dataclasses
    return dataclasses.replace(sc, material_palette=np.array(mats, np.int64).astype(np.int32), world_bvh=wn, actor_bvh=an, bvh_trigs=trigs,
                               name=sc.name + f"+{len(world)}+{len(actors)}tri")


FIXTURE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "benchmark_OpenCL_test.npz")
REFERENCE_SCENE = "/root/reference/benchmark/OpenCL_test/OpenCL_test"


ENTITY_FIXTURE = os.path.join(os.path.dirname(FIXTURE), "benchmark_OpenCL_test_entities.npz")


def cached_benchmark_scene(width: int = 1920, height: int = 1080, entities: bool = False) -> scenes.PackedScene:
    """The reference's `benchmark/OpenCL_test` scene (BASELINE.json configs[0]/[1]) as packed arrays.

    The committed fixture tests/golden/benchmark_OpenCL_test.npz is this module's conversion of the reference's data
    files (`OpenCL_test.octree2` + `.json`), written by `python -m chunkyclplugin_amd.octree2 --write-fixture` where
    /root/reference exists; the GPU box has no /root/reference and reads the fixture.  Raises FileNotFoundError when
    neither exists — callers must not skip silently."""
    if not os.path.exists(FIXTURE):
        if not os.path.exists(REFERENCE_SCENE + ".octree2"):
            raise FileNotFoundError("benchmark scene: neither " + REFERENCE_SCENE + ".octree2 nor " + FIXTURE)
        write_fixture()
    sc = scenes.load_scene(FIXTURE).with_view(width, height)
    if entities:  # the scene's 4 188 entities and 389 actors as box proxies in the two BVHs (table fixture: kinds, places, yaws)
        sc = with_entities(sc, np.load(ENTITY_FIXTURE)["table"])
    return sc


def write_fixture() -> str:
    scenes.save_scene(load_scene(REFERENCE_SCENE + ".octree2", REFERENCE_SCENE + ".json"), FIXTURE, compressed=True)
    js = json.loads(open(REFERENCE_SCENE + ".json", encoding="latin-1").read())
    chunks = js.get("chunkList", [])
    origin = (16.0 * min(c[0] for c in chunks), float(js.get("yMin", 0)), 16.0 * min(c[1] for c in chunks))
    np.savez_compressed(ENTITY_FIXTURE, table=entity_table(js, origin))
    return FIXTURE


if __name__ == "__main__":
    import sys
    if "--write-fixture" in sys.argv:
        print(write_fixture())
