"""Host-side mirror of the reference's device-binding layer, on top of the C ABI.

The reference's L2/L4 classes and their counterparts here:

=====================================================  ==========================================
reference (Java, JOCL)                                 this module (Python, ctypes -> libchunky_hip)
=====================================================  ==========================================
`RendererInstance.get()` (RendererInstance.java:23)    `RendererInstance.get(device)`
`ClSceneLoader` getters (ClSceneLoader.java:95-150)     `HipSceneLoader.load_packed(scene)` + getters
`ClCamera` (ClCamera.java:33-105)                       `HipPathTracingRenderer.set_camera`
`OpenClPathTracingRenderer.render` (:54-191)            `HipPathTracingRenderer.render`
`OpenClPreviewRenderer.render` (:47-115)                `HipPathTracingRenderer.preview`
=====================================================  ==========================================

What is NOT here: walking Chunky's `Scene` object (needs chunky-core; the Java side in
INTEGRATION.md does that and hands the same packed arrays to the same C entry points).
"""
from __future__ import annotations

import ctypes as C
import struct
from typing import Callable, Optional

import numpy as np

from . import native
from .native import check, ptr


class RendererInstance:
    """One GPU context per device index (the reference keeps a singleton for its one cl_device,
    RendererInstance.java:23-28; `clDevice` index from PersistentSettings :33)."""

    _instances = {}

    def __init__(self, device: int = 0):
        self.device = device
        self._h = C.c_void_p()
        check(native.lib().chunky_init(device, C.byref(self._h)))

    @classmethod
    def get(cls, device: int = 0) -> "RendererInstance":
        if device not in cls._instances:
            cls._instances[device] = RendererInstance(device)
        return cls._instances[device]

    @classmethod
    def group(cls, devices) -> "RendererInstance":
        """Several GPUs behind one context in this process (chunky_group_create): scenes are replicated, render targets
        are cut into 16 x 16-pixel blocks dealt round-robin to the members, `read()` gathers them on member 0.
        `devices` may repeat an index (members then share that GPU)."""
        self = cls.__new__(cls)
        self.device = int(devices[0])
        self.devices = [int(d) for d in devices]
        self._h = C.c_void_p()
        arr = (C.c_int * len(self.devices))(*self.devices)
        check(native.lib().chunky_group_create(arr, len(self.devices), C.byref(self._h)))
        return self

    def group_size(self) -> int:
        return native.lib().chunky_group_size(self._h)

    def peer_status(self) -> list:
        """chunky_group_peer_status: per member, how its share of a read-back reaches member 0 — native.PEER_LOCAL /
        PEER_DIRECT (xGMI) / PEER_STAGED, or a negated HIP error code when enabling peer access failed."""
        n = self.group_size()
        out = (C.c_int * n)()
        check(native.lib().chunky_group_peer_status(self._h, out, n))
        return list(out)

    TRANSPORT_NAMES = {native.TRANSPORT_PEER_COPY: "peer-copy", native.TRANSPORT_RCCL_SENDRECV: "rccl-sendrecv",
                       native.TRANSPORT_RCCL_REDUCE: "rccl-reduce"}

    def transport(self) -> dict:
        """chunky_group_transport: what carries the one exchange per read-back — {"transport": native.TRANSPORT_*, "name",
        "backend": "rccl" | "peer-copy", "detail": library / version / ranks, or the reason for the peer-copy fallback}."""
        t = C.c_int()
        buf = C.create_string_buffer(512)
        check(native.lib().chunky_group_transport(self._h, C.byref(t), buf, len(buf)))
        return {"transport": t.value, "name": self.TRANSPORT_NAMES.get(t.value, str(t.value)),
                "backend": "peer-copy" if t.value == native.TRANSPORT_PEER_COPY else "rccl",
                "detail": buf.value.decode("utf-8", "replace")}

    def set_transport(self, transport: int) -> None:
        """chunky_group_set_transport (ChunkyHipError with code E_STATE when it needs an RCCL communicator that is not there)."""
        check(native.lib().chunky_group_set_transport(self._h, int(transport)))

    @staticmethod
    def device_count() -> int:
        return native.lib().chunky_device_count()

    def device_name(self) -> str:
        buf = C.create_string_buffer(256)
        check(native.lib().chunky_device_name(self.device, buf, 256))
        return buf.value.decode()

    def selftest_math(self, which: int, a, b=None) -> np.ndarray:
        a = np.ascontiguousarray(a, np.float32)
        b = np.ascontiguousarray(a if b is None else b, np.float32)
        out = np.empty_like(a)
        check(native.lib().chunky_selftest_math(self._h, which, a.size, ptr(a), ptr(b), ptr(out)))
        return out

    def selftest_gamma_scan(self, curve: int, first_bits: int, count: int):
        """(values whose tone-map byte differs from the threshold table's, worst stray of the estimate) over `count` float bit patterns."""
        bad, worst = C.c_uint64(0), C.c_float(0)
        check(native.lib().chunky_selftest_gamma_scan(self._h, curve, first_bits, count, C.byref(bad), C.byref(worst)))
        return int(bad.value), float(worst.value)

    def close(self) -> None:
        if self._h:
            check(native.lib().chunky_shutdown(self._h))
            self._h = C.c_void_p()
            if RendererInstance._instances.get(self.device) is self:
                RendererInstance._instances.pop(self.device, None)


class HipSceneLoader:
    """Device copies of one packed scene — the buffer side of `ClSceneLoader`.

    `load_packed` takes the arrays Chunky's packers produce (here: `scenes.PackedScene`) and uploads
    them through the same entry points the JNI layer binds."""

    def __init__(self, instance: Optional[RendererInstance] = None):
        self.instance = instance or RendererInstance.get()
        self._h = C.c_void_p()
        check(native.lib().chunky_scene_create(self.instance._h, C.byref(self._h)))
        self.packed = None

    def load_packed(self, sc) -> bool:
        L = native.lib()
        i32 = lambda a: np.ascontiguousarray(a, np.int32)
        tree = i32(sc.octree)
        check(L.chunky_scene_set_octree(self._h, ptr(tree), tree.size, int(sc.octree_depth)))
        for kind, arr in ((native.PALETTE_BLOCK, sc.block_palette), (native.PALETTE_MATERIAL, sc.material_palette),
                          (native.PALETTE_AABB, sc.aabb_models), (native.PALETTE_QUAD, sc.quad_models),
                          (native.PALETTE_TRIG, sc.bvh_trigs)):
            a = i32(arr)
            check(L.chunky_scene_set_palette(self._h, kind, ptr(a), a.size))
        for which, arr in ((native.BVH_WORLD, sc.world_bvh), (native.BVH_ACTOR, sc.actor_bvh)):
            a = i32(arr)
            check(L.chunky_scene_set_bvh(self._h, which, ptr(a), a.size))
        atlas = np.ascontiguousarray(sc.atlas, np.uint8)
        layers, h, w, _ = atlas.shape
        check(L.chunky_scene_set_atlas(self._h, ptr(atlas), w, h, layers))
        sky = np.ascontiguousarray(sc.sky, np.uint8)
        check(L.chunky_scene_set_sky(self._h, ptr(sky), sky.shape[1], sky.shape[0], float(sc.sky_intensity)))
        sun = i32(sc.sun)
        check(L.chunky_scene_set_sun(self._h, ptr(sun)))
        self.packed = sc
        return True

    def emitters(self) -> np.ndarray:
        """The emitter list of the next-event-estimation extension: rows {x, y, z, level << 25 | block pointer}."""
        n = C.c_int32()
        check(native.lib().chunky_scene_emitters(self._h, None, 0, C.byref(n)))
        out = np.zeros((max(n.value, 1), 4), np.int32)
        check(native.lib().chunky_scene_emitters(self._h, ptr(out), n.value, C.byref(n)))
        return out[:n.value]

    def selftest_helpers(self, which: int, rows, tree: int = 1):
        """chunky_selftest_helpers: the device counterpart of reference helper `which` on rows of 32 floats; returns
        (rows of 12 floats, tree form used for row kind 14)."""
        rows = np.ascontiguousarray(rows, np.float32).reshape(-1, 32)
        out = np.zeros((len(rows), 12), np.float32)
        used = C.c_int32()
        check(native.lib().chunky_selftest_helpers(self._h, which, tree, len(rows), ptr(rows), ptr(out), C.byref(used)))
        return out, used.value

    def load_octree(self, tree_data, depth: int, block_mapping) -> None:
        """`ClSceneLoader.loadOctree` (ClSceneLoader.java:52-63): raw PackedOctree.treeData +
        blockMapping, remapped natively."""
        t = np.ascontiguousarray(tree_data, np.int32)
        m = np.ascontiguousarray(block_mapping, np.int32)
        check(native.lib().chunky_scene_load_octree(self._h, ptr(t), t.size, depth, ptr(m), m.size))

    def close(self) -> None:
        if self._h:
            check(native.lib().chunky_scene_destroy(self._h))
            self._h = C.c_void_p()


class HipPathTracingRenderer:
    """`OpenClPathTracingRenderer` for one image: ids, pass loop, read-back and merge."""

    ID = "ChunkyClRenderer"  # OpenClPathTracingRenderer.java:33-46 (getId/getName/getDescription)

    def __init__(self, scene_loader: HipSceneLoader, width: int, height: int):
        self.scene_loader = scene_loader
        self.width, self.height = width, height
        self._h = C.c_void_p()
        self.post_render: Optional[Callable[[], bool]] = None
        check(native.lib().chunky_render_create(scene_loader.instance._h, scene_loader._h, width, height,
                                                C.byref(self._h)))

    # --- Renderer interface -----------------------------------------------------------------
    def get_id(self) -> str:
        return self.ID

    def set_post_render(self, callback: Optional[Callable[[], bool]]) -> None:
        self.post_render = callback

    def auto_post_process(self) -> bool:
        return False  # OpenClPathTracingRenderer.java:197-200

    # --- camera / options -------------------------------------------------------------------
    def set_camera(self, projector_type: int, settings) -> None:
        s = np.ascontiguousarray(settings, np.float32)
        check(native.lib().chunky_render_set_camera(self._h, int(projector_type), ptr(s), s.size))

    def set_option(self, option: int, value) -> None:
        if option == native.OPT_EMITTER_SCALE:
            value = struct.unpack("<i", struct.pack("<f", float(value)))[0]
        check(native.lib().chunky_render_set_option(self._h, option, int(value)))

    def set_shard(self, rank: int, world: int, tile: int = 256) -> None:
        check(native.lib().chunky_render_set_shard(self._h, rank, world, tile))

    def set_device_buffer(self, device_ptr: Optional[int]) -> None:
        check(native.lib().chunky_render_set_device_buffer(self._h, device_ptr))

    def device_buffer(self) -> int:
        p = C.c_void_p()
        check(native.lib().chunky_render_device_buffer(self._h, C.byref(p)))
        return p.value

    # --- passes -------------------------------------------------------------------------------
    def reset(self) -> None:
        check(native.lib().chunky_render_reset(self._h))

    def render_passes(self, seeds, first_buffer_spp: int = 0, sync: bool = True) -> None:
        s = np.ascontiguousarray(seeds, np.int32)
        check(native.lib().chunky_render_passes(self._h, ptr(s), s.size, first_buffer_spp))
        if sync:
            self.sync()

    def sync(self) -> None:
        check(native.lib().chunky_render_sync(self._h))

    def gather(self) -> None:
        """The read-back exchange without the copy to the host (a group: member 0's device buffer then holds the image)."""
        check(native.lib().chunky_render_gather(self._h))

    def read(self, out: Optional[np.ndarray] = None) -> np.ndarray:
        """chunky_render_read (on a group: the one exchange, then the copy); `out` = a float32 array to fill — e.g. pinned memory."""
        if out is None:
            out = np.empty(self.width * self.height * 3, np.float32)
        assert out.dtype == np.float32 and out.size == self.width * self.height * 3 and out.flags["C_CONTIGUOUS"]
        check(native.lib().chunky_render_read(self._h, ptr(out), out.size))
        return out

    def kernel_time(self):
        ms, n = C.c_float(), C.c_int()
        check(native.lib().chunky_render_kernel_time(self._h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def kernel_info(self) -> dict:
        """The kernel instantiation the last launch ran: tree form, lanes per pixel, entity-BVH phases, workgroups."""
        out = np.zeros(8, np.int32)
        check(native.lib().chunky_render_kernel_info(self._h, ptr(out)))
        return {"tree": int(out[0]), "group": int(out[1]), "bvh": bool(out[2]), "blocks": int(out[3]), "pool": int(out[4]),
                "ext": bool(out[5]), "passes_per_launch": int(out[6]), "sorted": bool(out[7])}

    def phase_stats(self, reset: bool = True) -> dict:
        out = np.zeros(24, np.uint64)
        check(native.lib().chunky_render_phase_stats(self._h, ptr(out), 1 if reset else 0))
        o = out[:9].reshape(3, 3)
        d = {name: {"execs": int(o[i, 0]), "lanes": int(o[i, 1]), "cycles": int(o[i, 2])}
             for i, name in enumerate(("march", "block", "shade"))}
        d["waves"] = {"life_sum": int(out[9]), "life_max": int(out[10]), "n": int(out[11])}
        d["handover"] = {"execs": int(out[12]), "cycles": int(out[13])}
        d["parts"] = {name: int(out[14 + i]) for i, name in enumerate(
            ("sky", "sampling", "trace_setup", "deposit", "fold", "open_pixel", "hand_out", "new_sample"))}
        # render_pool's sorted instantiation: the model blocks' phase ("block" is then the full cubes alone)
        d["model"] = {"execs": int(out[22]) >> 40, "lanes": int(out[22]) & ((1 << 40) - 1), "cycles": int(out[23])}
        return d

    def preview(self) -> np.ndarray:
        out = np.empty(self.width * self.height, np.int32)
        check(native.lib().chunky_render_preview(self._h, ptr(out)))
        return out

    def trace_records(self, seed: int, gids):
        g = np.ascontiguousarray(gids, np.int32)
        rec = np.zeros((g.size, native.MAX_TRACES), native.HIT_DTYPE)
        cnt = np.zeros(g.size, np.int32)
        rad = np.zeros((g.size, 3), np.float32)
        check(native.lib().chunky_render_trace_records(self._h, int(seed), ptr(g), g.size, ptr(rec), ptr(cnt), ptr(rad)))
        return rec, cnt, rad

    def render(self, sample_buffer: np.ndarray, scene_spp: int, target_spp: int, merge_interval: int = 1024) -> int:
        """The pass loop of OpenClPathTracingRenderer.render (:95-184) run natively; merges into the
        caller's double sample buffer and returns the new scene.spp."""
        assert sample_buffer.dtype == np.float64 and sample_buffer.size == self.width * self.height * 3
        spp = C.c_int32(scene_spp)
        cb = native.POST_RENDER_FN((lambda _u: 1 if self.post_render() else 0) if self.post_render else 0)
        rc = native.lib().chunky_render_run(self._h, ptr(sample_buffer), C.byref(spp), target_spp, merge_interval, cb, None)
        if rc not in (0, native.E_ABORTED):
            check(rc)
        return spp.value

    def render_ex(self, sample_buffer: np.ndarray, scene_spp: int, target_spp: int, merge_interval: int = 1024,
                  progress=None, merged=None, save_event=None, regenerate_camera=None, poll_gate=None) -> int:
        """chunky_render_run_ex: the same loop with the reference's other hooks — `progress(spp)` after every launch
        (scene.spp, :144), `merged(spp)` after every merge (:172-177), `save_event(spp) -> bool` (isSaveEvent, :150) and
        `regenerate_camera()` between launches (:146-148).  `save_event` may return 2 for "merge now, no extra poll"
        (scene.shouldFinalizeBuffer()); `poll_gate() -> bool` gates the timed poll only (`!manager.shouldFinalize()`, :154).
        Returns the new scene.spp."""
        assert sample_buffer.dtype == np.float64 and sample_buffer.size == self.width * self.height * 3
        spp = C.c_int32(scene_spp)
        cb = native.RunCallbacks(
            C.sizeof(native.RunCallbacks), native.POST_RENDER_FN((lambda _u: 1 if self.post_render() else 0) if self.post_render else 0),
            native.PROGRESS_FN((lambda _u, s: progress(s)) if progress else 0),
            native.PROGRESS_FN((lambda _u, s: merged(s)) if merged else 0),
            native.SAVE_EVENT_FN((lambda _u, s: int(save_event(s))) if save_event else 0),
            native.REGEN_FN((lambda _u: regenerate_camera()) if regenerate_camera else 0), None,
            native.POST_RENDER_FN((lambda _u: 1 if poll_gate() else 0) if poll_gate else 0))
        rc = native.lib().chunky_render_run_ex(self._h, ptr(sample_buffer), C.byref(spp), target_spp, merge_interval, C.byref(cb))
        if rc not in (0, native.E_ABORTED):
            check(rc)
        return spp.value

    def close(self) -> None:
        if self._h:
            check(native.lib().chunky_render_destroy(self._h))
            self._h = C.c_void_p()


class HipPostProcessingFilter:
    """GPU tone mapping — GpuPostProcessingFilter / ImposterCombinationGpuPostProcessingFilter
    (tonemap/GpuPostProcessingFilter.java:14-82): takes the name, description and id of the Chunky
    filter it stands in for (ChunkyCl.java:60-63: GAMMA, TONEMAP1, TONEMAP2 = ACES, TONEMAP3 = HABLE)
    and runs the `filter` kernel (tonemap/include/post_processing_filter.cl:5-51) in process_frame."""

    GAMMA, TONEMAP1, ACES, HABLE = 0, 1, 2, 3
    IMPOSTERS = {"GAMMA": GAMMA, "TONEMAP1": TONEMAP1, "TONEMAP2": ACES, "TONEMAP3": HABLE}

    def __init__(self, filter_id: str, instance: Optional[RendererInstance] = None, name: str = None, description: str = None):
        if filter_id not in self.IMPOSTERS:
            raise ValueError(f"no GPU filter for post-processing id {filter_id!r}")
        self.instance = instance or RendererInstance.get()
        self.id, self.type = filter_id, self.IMPOSTERS[filter_id]
        self.name = name or filter_id
        self.description = description or filter_id

    def get_id(self) -> str:
        return self.id

    def get_name(self) -> str:
        return self.name

    def get_description(self) -> str:
        return self.description

    def process_frame(self, width: int, height: int, input_samples: np.ndarray, output_argb: np.ndarray, exposure: float) -> None:
        """processFrame (GpuPostProcessingFilter.java:40-65): `input_samples` = width*height*3 doubles, `output_argb` =
        width*height int32 (BitmapImage.data); blocking."""
        if input_samples.dtype != np.float64 or input_samples.size != 3 * width * height or not input_samples.flags.c_contiguous:
            raise ValueError("input must be a contiguous float64 array of width*height*3 samples")
        if output_argb.dtype not in (np.int32, np.uint32) or output_argb.size != width * height or not output_argb.flags.c_contiguous:
            raise ValueError("output must be a contiguous int32 array of width*height pixels")
        check(native.lib().chunky_filter_frame(self.instance._h, width, height, float(exposure), ptr(input_samples),
                                               ptr(output_argb), self.type))

    def process_device(self, n_pixels: int, exposure: float, d_input: int, d_argb: int, repeat: int = 1) -> float:
        """The same kernel on device buffers; returns the mean kernel time in ms (HIP events)."""
        ms = C.c_float()
        check(native.lib().chunky_filter_frame_device(self.instance._h, n_pixels, float(np.float32(exposure)), C.c_void_p(d_input),
                                                      C.c_void_p(d_argb), self.type, repeat, C.byref(ms)))
        return ms.value
