"""The kernel instantiations that bench.py / tools/config_bench.py TIME, compared with the oracle.

`launch_render` picks the lookup form (TREE), the lanes per pixel (G) and the entity-BVH phases from the scene, the shard
size and the pass count, so a small test scene never runs the kernel the benchmark runs.  These tests render the
benchmark workloads themselves — full 1920x1080 views, 32 or more passes per launch — assert through
chunky_render_kernel_info that the instantiation is the timed one, and compare whole image rows with the C restatement
(oracle/port.c `port_render_gids`), bit for bit.  Reference loops covered: K/rayTracer.cl:93-112, K/octree.h:66-106,
K/bvh.h:22-113."""
import os

import numpy as np
import pytest

from oracle import binding

from chunkyclplugin_amd import native, parallel, scenes
from chunkyclplugin_amd.renderer import HipPathTracingRenderer, HipSceneLoader

pytestmark = pytest.mark.gpu
THREADS = binding.usable_threads()
ROWS = (7, 101, 263, 411, 540, 688, 799, 931, 1003, 1079)


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def make(gpu_instance, sc):
    loader = HipSceneLoader(gpu_instance)
    loader.load_packed(sc)
    r = HipPathTracingRenderer(loader, sc.width, sc.height)
    r.set_camera(sc.projector_type, sc.camera)
    return loader, r


def row_gids(sc, rows=ROWS):
    rows = [min(y, sc.height - 1) for y in rows]
    return np.concatenate([np.arange(y * sc.width, (y + 1) * sc.width) for y in rows]).astype(np.int32)


def compare_rows(r, port, sc, seeds, gids, what, first=0):
    got = r.read().reshape(-1, 3)[gids]
    want = port.render_gids(sc, seeds, gids, first_spp=first, threads=THREADS).reshape(-1, 3)[gids]
    same = (bits(got) == bits(want)).all(axis=1)
    if not same.all():
        rel = np.abs(got.astype(np.float64) - want) / np.maximum(np.abs(want), 1e-6)
        pytest.fail(f"{what}: {int((~same).sum())} of {len(gids)} pixels differ from the oracle "
                    f"(max rel err {np.nanmax(rel):.3e}, first gid {int(gids[np.argmin(same)])})")
    assert np.isfinite(got).all() and got.max() > 0, what


@pytest.fixture(scope="module")
def outdoor():
    return scenes.cached_outdoor_world(chunks=32, height=256)  # BASELINE configs[2], 1920x1080: what bench.py renders


def test_headline_kernel_256_passes(gpu_instance, port, outdoor):
    """bench.py's step: one 256-pass launch of render_pool<17, 64> (+ fold_kernel) over the whole 1080p image."""
    sc = outdoor
    seeds = native.java_random_ints(256)
    loader, r = make(gpu_instance, sc)
    r.render_passes(seeds)
    info = r.kernel_info()
    assert (info["tree"], info["pool"], info["bvh"]) == (17, 64, False), info
    assert info["blocks"] >= 256 * 4
    compare_rows(r, port, sc, seeds, row_gids(sc, ROWS[::3]), "outdoor 256 passes")
    # the second step of the bench continues the running mean at bufferSpp = 256 (K/rayTracer.cl:109-112)
    more = native.java_random_ints(288)[256:]
    r.render_passes(more, first_buffer_spp=256)
    g = row_gids(sc, (263, 931))
    got = r.read().reshape(-1, 3)[g]
    want = port.render_gids(sc, seeds, g, threads=THREADS)
    want = port.render_gids(sc, more, g, first_spp=256, res=want, threads=THREADS).reshape(-1, 3)[g]
    np.testing.assert_array_equal(bits(got), bits(want))
    r.close()
    loader.close()


@pytest.mark.parametrize("world,passes,group,variant", [(1, 32, 0, 0), (4, 32, 0, 0), (8, 64, 0, 0), (2, 48, 0, 0),
                                                        (1, 32, 8, 8), (4, 32, 16, 8), (8, 64, 32, 8)])
def test_outdoor_shard_shares(gpu_instance, port, outdoor, world, passes, group, variant):
    """The tile split of bench.py --gpus N (rank 1 of N, 256-pixel tiles) with the pool kernel (a work item is a sample,
    so every share runs the same instantiation), and with the grouped kernel (variant 8), where a quarter of the image
    runs 16 lanes per pixel and an eighth 32."""
    sc = outdoor
    seeds = native.java_random_ints(passes)
    loader, r = make(gpu_instance, sc)
    r.set_option(native.OPT_KERNEL, variant)
    rank = 1 if world > 1 else 0
    r.set_shard(rank, world, 256)
    r.render_passes(seeds)
    info = r.kernel_info()
    if variant == 0:
        assert (info["tree"], info["pool"], info["bvh"]) == (17, 64, False), info
    else:
        assert (info["tree"], info["group"], info["bvh"], info["pool"]) == (-1, group, False, -1), info   # the fallback: generic tree form
    own = parallel.owned_gids(sc.width * sc.height, rank, world, 256)
    rows = row_gids(sc)
    mine = np.intersect1d(rows, own)
    assert mine.size >= rows.size // world - 256
    compare_rows(r, port, sc, seeds, mine, f"outdoor share 1/{world}")
    others = np.setdiff1d(rows, own)
    assert not r.read().reshape(-1, 3)[others].any()          # pixels of other ranks stay zero (the reduce adds them)
    r.close()
    loader.close()


@pytest.mark.parametrize("world,rank,passes", [(2, 1, 32), (8, 5, 64), (3, 0, 32)])
def test_outdoor_block_shards(gpu_instance, port, outdoor, world, rank, passes):
    """The split bench.py --gpus N uses (chunky_render_set_shard with tile 0): the image's 16 x 16 blocks dealt round-robin,
    rendered by the pool kernel in the same tile shape as the whole image.  The rank's pixels of whole image rows (the half-
    padded bottom block row of 1080 lines included) against the oracle, everybody else's stay zero, and the ranks' pixel
    sets partition the image; the grouped kernel (which has no block mapping) renders the same share from a pixel list."""
    sc = outdoor
    n = sc.width * sc.height
    seeds = native.java_random_ints(passes)
    loader, r = make(gpu_instance, sc)
    r.set_shard(rank, world, 0)
    r.render_passes(seeds)
    info = r.kernel_info()
    assert (info["tree"], info["pool"], info["bvh"]) == (17, 64, False), info
    own = parallel.owned_gids(n, rank, world, 0, sc.width)
    rows = row_gids(sc, ROWS + (1072,))
    mine = np.intersect1d(rows, own)
    assert abs(mine.size - rows.size / world) < 0.02 * rows.size + 32
    compare_rows(r, port, sc, seeds, mine, f"outdoor block share {rank}/{world}")
    img = r.read().reshape(-1, 3)
    mask = np.ones(n, bool)
    mask[own] = False
    assert not img[mask].any()                     # every pixel of the other ranks stays zero (the reduce adds them)
    assert np.count_nonzero(img[own].any(axis=1)) > 0.9 * own.size
    r.set_option(native.OPT_KERNEL, 8)
    r.reset()
    r.render_passes(seeds[:2])
    assert r.kernel_info()["pool"] < 0
    compare_rows(r, port, sc, seeds[:2], mine, f"outdoor block share {rank}/{world}, round 1's kernel")
    assert not r.read().reshape(-1, 3)[mask].any()
    r.close()
    loader.close()


@pytest.mark.parametrize("world,rank,passes,cap", [(8, 3, 700, 1024), (1, 0, 300, 342), (2, 1, 600, 685)])
def test_launches_longer_than_the_argument_segment(gpu_instance, port, outdoor, world, rank, passes, cap):
    """render_pool carries up to 1024 passes per launch when the staged samples fit 8 GiB (a share of the image on several GPUs
    pays the end-of-launch tail once instead of four times); beyond 256 passes the seeds travel in device memory.  One call of
    `passes` passes is then ONE launch, and the image is the oracle's."""
    sc = outdoor
    seeds = native.java_random_ints(passes)
    loader, r = make(gpu_instance, sc)
    r.set_shard(rank, world, 0)
    r.kernel_time()
    r.render_passes(seeds)
    info = r.kernel_info()
    assert (info["tree"], info["pool"], info["passes_per_launch"]) == (17, 64, cap), info
    assert r.kernel_time()[1] == 1                                    # one launch
    own = parallel.owned_gids(sc.width * sc.height, rank, world, 0, sc.width)
    mine = np.intersect1d(row_gids(sc, (411, 1003))[::3], own)
    compare_rows(r, port, sc, seeds, mine, f"{passes} passes in one launch, share {rank}/{world}")
    # a second long launch continues the running mean (and reuses the seed buffer behind the first)
    more = native.java_random_ints(passes + 290)[passes:]
    r.render_passes(more, first_buffer_spp=passes)
    g = mine[::7]
    got = r.read().reshape(-1, 3)[g]
    want = port.render_gids(sc, seeds, g, threads=THREADS)
    want = port.render_gids(sc, more, g, first_spp=passes, res=want, threads=THREADS).reshape(-1, 3)[g]
    np.testing.assert_array_equal(bits(got), bits(want))
    # the fallback kernels keep to what the argument segment holds
    r.set_option(native.OPT_KERNEL, 8)
    r.reset()
    r.kernel_time()
    r.render_passes(seeds[:300])
    assert r.kernel_info()["pool"] < 0 and r.kernel_info()["passes_per_launch"] == 256 and r.kernel_time()[1] == 2
    compare_rows(r, port, sc, seeds[:300], mine[::5], "300 passes, round 1's kernel: two launches")
    r.close()
    loader.close()


def test_city_kernel(gpu_instance, port):
    """BASELINE configs[1]: the reference's benchmark octree (depth 10) at 1920x1080 — render_pool<17, 64> (a 7-bit dense top over one level)."""
    from chunkyclplugin_amd import octree2
    sc = octree2.cached_benchmark_scene(1920, 1080)   # raises when the fixture is missing: never skipped silently
    seeds = native.java_random_ints(64)
    loader, r = make(gpu_instance, sc)
    r.render_passes(seeds)
    info = r.kernel_info()
    assert (info["tree"], info["pool"], info["bvh"]) == (17, 64, False), info  # depth 10 = a 7-bit dense top over one 3-bit level
    compare_rows(r, port, sc, seeds, row_gids(sc), "city 64 passes")
    r.set_shard(3, 8, 256)
    r.reset()
    r.render_passes(seeds)
    info = r.kernel_info()
    assert (info["tree"], info["pool"]) == (17, 64), info
    own = parallel.owned_gids(sc.width * sc.height, 3, 8, 256)
    compare_rows(r, port, sc, seeds, np.intersect1d(row_gids(sc), own), "city share 1/8")
    r.close()
    loader.close()


def test_city_with_its_entities(gpu_instance, port):
    """configs[1] with the scene's own 4 188 entities and 389 actors (box proxies, octree2.with_entities) in the world and
    actor BVHs: render_pool<17, ., bvh> on the depth-10 octree, whole rows against the oracle; and a close-up view in
    which the entities fill the picture."""
    from chunkyclplugin_amd import octree2
    sc = octree2.cached_benchmark_scene(1920, 1080, entities=True)
    seeds = native.java_random_ints(16)
    loader, r = make(gpu_instance, sc)
    r.render_passes(seeds)
    info = r.kernel_info()
    assert (info["tree"], info["bvh"]) == (17, True) and info["pool"] in (16, 32), info
    compare_rows(r, port, sc, seeds, row_gids(sc, ROWS[::3]), "city + entities")
    r.close()
    loader.close()
    table = np.load(octree2.ENTITY_FIXTURE)["table"]
    e = table[table[:, 0] == 0][len(table) // 9]              # a painting somewhere in the city
    near = sc.with_view(320, 200, camera=scenes.look_at_camera((e[1] + 4.0, e[2] + 1.5, e[3] + 3.0), (e[1], e[2], e[3]), 60.0))
    loader, r = make(gpu_instance, near)
    r.render_passes(seeds[:6])
    want = port.render_passes(near, seeds[:6], threads=THREADS)
    np.testing.assert_array_equal(bits(r.read()), bits(want))
    bare = octree2.cached_benchmark_scene(320, 200).with_view(320, 200, camera=near.camera)
    assert not np.array_equal(bits(port.render_passes(bare, seeds[:2], threads=THREADS)), bits(port.render_passes(near, seeds[:2], threads=THREADS)))
    r.close()
    loader.close()


def test_config0_plumbing_image(gpu_instance, port):
    """BASELINE configs[0] — benchmark/OpenCL_test at 256x256, 16 spp (recorded on the CPU path by tools/config0_cpu.py,
    profiles/r05_config0_cpu.json): the whole image from the HIP path equals the CPU path's, bit for bit."""
    from chunkyclplugin_amd import octree2
    sc = octree2.cached_benchmark_scene(256, 256)
    seeds = native.java_random_ints(16)
    loader, r = make(gpu_instance, sc)
    r.render_passes(seeds)
    got = r.read()
    want = port.render_passes(sc, seeds, threads=THREADS)
    np.testing.assert_array_equal(bits(got), bits(want))
    r.close()
    loader.close()


def test_indoor_kernel(gpu_instance, port):
    """BASELINE configs[3]: the emitter-lit room of tools/config_bench.py (sun flag 0), 1920x1080, 32 passes."""
    sc = scenes.indoor_room(size=64, width=1920, img_height=1080)
    seeds = native.java_random_ints(32)
    loader, r = make(gpu_instance, sc)
    r.render_passes(seeds)
    info = r.kernel_info()
    assert (info["tree"], info["pool"], info["bvh"]) == (17, 64, False), info
    compare_rows(r, port, sc, seeds, row_gids(sc, ROWS[1::2]), "indoor 32 passes")
    r.close()
    loader.close()


@pytest.fixture(scope="module")
def entity_world(outdoor):
    # BASELINE configs[4] as tools/config_bench.py builds it: 100 000 world + 5 000 actor triangles
    return scenes.add_entities(outdoor, 100000, seed=11, actor_tris=5000, region=((40, 90, 40), (470, 170, 470)))


@pytest.mark.parametrize("world,passes,group,variant", [(1, 16, 0, 0), (4, 32, 0, 0), (1, 16, 8, 8), (4, 32, 16, 8)])
def test_entity_kernels(gpu_instance, port, entity_world, world, passes, group, variant):
    """The 100 k-triangle world (BVHs of height ~17, both BVHs walked): render_pool<17, 16 | 32, bvh> on the aligned
    node / triangle records (default), and the fallback render_waves<-1, 8 | 16, true> on the packed arrays (variant 8)."""
    sc = entity_world
    seeds = native.java_random_ints(passes)
    loader, r = make(gpu_instance, sc)
    r.set_option(native.OPT_KERNEL, variant)
    rank = world - 1
    r.set_shard(rank, world, 256)
    r.render_passes(seeds)
    info = r.kernel_info()
    if variant == 0:
        assert (info["tree"], info["bvh"]) == (17, True) and info["pool"] in (16, 32), info
    else:
        assert (info["tree"], info["group"], info["bvh"], info["pool"]) == (-1, group, True, -1), info
    own = parallel.owned_gids(sc.width * sc.height, rank, world, 256)
    mine = np.intersect1d(row_gids(sc, (101, 411, 540, 799, 1003)), own)
    compare_rows(r, port, sc, seeds, mine, f"entities share 1/{world}")
    r.close()
    loader.close()


def test_entity_config_at_its_stated_size(gpu_instance, port, entity_world):
    """BASELINE configs[4] as stated — 3840 x 2160, tiles split over 8 GPUs: what one GPU of the eight renders (rank 5's
    16 x 16 blocks), on this one GPU, against the oracle on that rank's pixels of five image rows; nobody else's pixels
    are touched."""
    sc = entity_world.with_view(3840, 2160)
    seeds = native.java_random_ints(8)
    loader, r = make(gpu_instance, sc)
    r.set_shard(5, 8, 0)
    r.render_passes(seeds)
    info = r.kernel_info()
    assert (info["tree"], info["bvh"]) == (17, True) and info["pool"] in (16, 32), info
    own = parallel.owned_gids(sc.width * sc.height, 5, 8, 0, sc.width)
    assert abs(own.size - sc.width * sc.height / 8) < 4096
    mine = np.intersect1d(row_gids(sc, (203, 822, 1080, 1599, 2007)), own)
    compare_rows(r, port, sc, seeds, mine, "entities 4K share 5/8")
    mask = np.ones(sc.width * sc.height, bool)
    mask[own] = False
    assert not r.read().reshape(-1, 3)[mask].any()
    r.close()
    loader.close()


def test_entity_trace_records(gpu_instance, port, entity_world):
    """Every closestIntersect of one sample on ~100 pixels of the 100 k-triangle world: hit flags and block indices exact,
    every float field bit-identical (octree + world BVH + actor BVH, main and shadow traces)."""
    sc = entity_world
    loader, r = make(gpu_instance, sc)
    gids = np.arange(1920 * 300 + 5, 1920 * 1080, 1920 * 780 // 100 + 13, dtype=np.int32)[:100]
    seed = int(native.java_random_ints(3)[2])
    rec, cnt, rad = r.trace_records(seed, gids)
    import dataclasses
    bare = dataclasses.replace(sc, world_bvh=scenes.empty_bvh(), actor_bvh=scenes.empty_bvh())  # the same world without entities
    touched = 0
    for i, gid in enumerate(gids):
        want, wrad = port.trace_records(sc, seed, int(gid))
        n = int(cnt[i])
        assert n == len(want), (gid, n, len(want))
        got = rec[i, :n]
        assert got["hit"].tolist() == want["hit"].tolist(), gid
        assert got["material"].tolist() == want["material"].tolist(), gid
        for f in ("distance", "normal", "color", "emittance"):
            np.testing.assert_array_equal(bits(got[f]), bits(want[f]), err_msg=f"gid {gid} {f}")
        hit = want["hit"] == 1
        np.testing.assert_array_equal(bits(got["point"][hit]), bits(want["point"][hit]))
        np.testing.assert_array_equal(bits(rad[i]), bits(wrad))
        w0, _ = port.trace_records(bare, seed, int(gid))
        touched += int(len(w0) != len(want) or not np.array_equal(bits(w0["distance"]), bits(want["distance"])))
    assert touched >= 5, f"only {touched} of {len(gids)} sampled paths meet an entity: the sample does not exercise the BVH path"
    r.close()
    loader.close()


def test_million_triangle_world(gpu_instance, port):
    """BASELINE configs[4] at the upper end of its range (10^5 - 10^6 triangles): 1 000 000 world + 5 000
    actor triangles — BVHs of half a million inner nodes, 122 MB of node / triangle records behind 32-bit offsets — whole rows
    against the oracle.  (The scene comes from .scene_cache when present; building its BVH takes about a minute otherwise.)"""
    sc = scenes.cached_entity_world(1000000)
    seeds = native.java_random_ints(2)
    loader, r = make(gpu_instance, sc)
    r.render_passes(seeds)
    info = r.kernel_info()
    assert (info["tree"], info["bvh"]) == (17, True) and info["pool"] in (16, 32), info
    compare_rows(r, port, sc, seeds, row_gids(sc, (411, 799)), "10^6 triangles")
    r.close()
    loader.close()
