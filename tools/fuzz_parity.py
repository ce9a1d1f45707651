#!/usr/bin/env python3
"""tools/fuzz_parity.py [iterations] [first seed] — randomised parity sweep on the GPU box: random small worlds, views, pass
counts, kernel variants, shards, render-loop constants and (a third of the time) the experimental light-transport options,
each rendered through the C ABI and compared with the oracle bit for bit.  Not part of the test suite (it is a search, not a
check): prints one line per failure and a summary; exit code 1 if anything differed."""
import dataclasses
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
# the fuzz moves the BVH records about (CHUNKY_BVH_LAYOUT) and sends a one-member group's own blocks through RCCL (CHUNKY_GROUP_SELF_EXCHANGE):
# rig variables of the -DCHUNKY_TUNING build only, which is what it therefore loads (set before the binding is imported)
if "CHUNKY_HIP_LIB" not in os.environ:
    os.environ["CHUNKY_HIP_LIB"] = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "chunkyclplugin_amd", "libchunky_hip_tuning.so")
from chunkyclplugin_amd import native, parallel, scenes  # noqa: E402
from chunkyclplugin_amd.renderer import HipPathTracingRenderer, HipSceneLoader, RendererInstance  # noqa: E402
from oracle import binding  # noqa: E402
from oracle.binding import PortCull, PortExt, PortOptions  # noqa: E402

VARIANTS = [0, 0, 0, 256, 256, 512, 1, 2, 3, 64, 128, 8, 9, 8 | 16, 8 | 32, 8 | 48]  # (256 / 512: render_pool's sorted block tests forced on / off)


def main():
    n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    inst = RendererInstance.get(0)
    groups = {n: RendererInstance.group([0] * n) for n in (2, 3)}   # several members behind one context (chunky_group_create), sharing the GPU
    # ... and a one-member group, which owns a real RCCL communicator: with CHUNKY_GROUP_SELF_EXCHANGE its own blocks go through the
    # read-back exchange (packed, sent to itself, received, scattered — or reduced in place), by a transport drawn per configuration
    os.environ["CHUNKY_GROUP_SELF_EXCHANGE"] = "1"
    groups[1] = RendererInstance.group([0])
    rccl = groups[1].transport()["backend"] == "rccl"
    print("fuzz: one-member group:", groups[1].transport(), flush=True)
    transports_used = {}
    port = binding.port()
    bad = 0
    for it in range(n_iter):
        rng = np.random.default_rng(seed0 + it)
        size = int(rng.choice([16, 32, 48]))
        ents = int(rng.choice([0, 0, 24, 120]))
        w, h = int(rng.integers(5, 90)), int(rng.integers(3, 60))
        if rng.random() < 0.15:  # a few hundred tiles: every XCD range of render_pool's sample queue has work
            w, h = int(rng.integers(150, 400)), int(rng.integers(100, 260))
        sc = scenes.tiny_scene(seed=int(rng.integers(1, 10 ** 6)), size=size, width=w, height=h, entities=ents, sun_flag=bool(rng.random() < 0.7))
        if rng.random() < 0.3:   # look from inside / from far outside the world
            S = float(1 << sc.octree_depth)
            eye = rng.uniform(-0.5 * S, 1.5 * S, 3)
            sc = dataclasses.replace(sc, camera=scenes.look_at_camera(tuple(eye), tuple(rng.uniform(0.2 * S, 0.8 * S, 3)), float(rng.uniform(30, 110))))
        ext = {}
        if rng.random() < 0.33:
            m = np.asarray(sc.material_palette).copy().reshape(-1, 6)
            m[:, 5] = rng.integers(0, 256, len(m)) | (rng.integers(0, 256, len(m)) << 8) | (rng.integers(0, 256, len(m)) << 16)
            m[rng.random(len(m)) < 0.4, 5] = 0
            sc = dataclasses.replace(sc, material_palette=m.reshape(-1).astype(np.int32))
            ext = {"bsdf": int(rng.integers(0, 2)), "nee": int(rng.integers(0, 2)), "sun_sampling": int(rng.integers(-1, 2)), "emitters": int(rng.integers(0, 2))}
            if ext == {"bsdf": 0, "nee": 0, "sun_sampling": -1, "emitters": 1}:
                ext["bsdf"] = 1
        variant = 0 if ext else int(rng.choice(VARIANTS))
        passes = int(rng.choice([1, 2, 3, 7, 16, 33, 70]))
        if w * h < 2000 and rng.random() < 0.08:
            passes = int(rng.choice([257, 300, 700, 1024, 1100]))  # launches longer than the argument segment holds seeds for (pool kernel), or cut in several
        first = int(rng.choice([0, 0, 5, 1000]))
        draw, depth, scale = int(rng.choice([256, 256, 40, 3])), int(rng.choice([5, 5, 1, 2, 9])), float(rng.choice([13.0, 13.0, 0.0, 2.5]))
        world = int(rng.choice([1, 1, 2, 3, 8]))
        rank, tile = int(rng.integers(0, world)), int(rng.choice([256, 64, 100]))
        if rng.random() < (0.5 if variant in (0, 64, 128) else 0.25):
            tile = 0  # 16 x 16 blocks: mapped by the pool kernel itself, handed to the other kernels as a pixel list
        seeds = native.java_random_ints(passes, seed=int(rng.integers(0, 10 ** 6)))
        # a fifth of the pool-kernel cases run on a multi-member group: its members split the share again, the read-back gathers
        # (any kernel variant: members hold block shares, which the fallback kernels render from a pixel list; a large draw depth
        # sends the pool kernel's cases to the fallback too)
        on_group = int(rng.choice([1, 1, 2, 3])) if rng.random() < 0.3 else 0
        if on_group == 1:
            t = int(rng.choice([native.TRANSPORT_RCCL_SENDRECV, native.TRANSPORT_RCCL_REDUCE, native.TRANSPORT_PEER_COPY])) if rccl else native.TRANSPORT_PEER_COPY
            groups[1].set_transport(t)
            transports_used[t] = transports_used.get(t, 0) + 1
        if not ext and rng.random() < 0.05:
            draw = 70000
        # entity-BVH placement (addresses only) and the behind-the-ray cull (an extension with its own oracle mode)
        layout = str(rng.choice(["0,0", "0,0", "3,4", "64,8", "4096,32"]))
        os.environ["CHUNKY_BVH_LAYOUT"] = layout
        cull = int(ents > 0 and rng.random() < 0.3)
        if os.environ.get("FUZZ_VERBOSE"):
            import time
            print(f"it={it} view={w}x{h} ents={ents} variant={variant} passes={passes} first={first} draw={draw} depth={depth} shard={rank}/{world}/{tile} "
                  f"group={on_group} ext={ext} layout={layout} cull={cull} t={time.time():.2f}", flush=True)
        loader = HipSceneLoader(groups[on_group] if on_group else inst)
        loader.load_packed(sc)
        r = HipPathTracingRenderer(loader, w, h)
        r.set_camera(sc.projector_type, sc.camera)
        r.set_option(native.OPT_KERNEL, variant)
        r.set_option(native.OPT_DRAW_DEPTH, draw)
        r.set_option(native.OPT_MAX_DEPTH, depth)
        r.set_option(native.OPT_EMITTER_SCALE, scale)
        for k, v in ext.items():
            r.set_option({"sun_sampling": native.OPT_SUN_SAMPLING, "emitters": native.OPT_EMITTERS, "bsdf": native.OPT_BSDF, "nee": native.OPT_EMITTER_NEE}[k], v)
        r.set_option(native.OPT_BVH_CULL_BEHIND, cull)
        r.set_shard(rank, world, tile)
        r.render_passes(seeds, first_buffer_spp=first)
        got = r.read()
        own = parallel.owned_gids(w * h, rank, world, tile, w)
        want = np.zeros(3 * w * h, np.float32)
        with PortOptions(port, draw, depth, scale), PortCull(port, on=bool(cull)):
            if ext:
                with PortExt(port, sc, **ext):
                    port.render_gids(sc, seeds, own, first_spp=first, res=want)
            else:
                port.render_gids(sc, seeds, own, first_spp=first, res=want)
        if not np.array_equal(got.view(np.uint32), want.view(np.uint32)):
            bad += 1
            diff = np.nonzero(got.view(np.uint32) != want.view(np.uint32))[0]
            print(f"FAIL it={it} seed={seed0 + it} size={size} ents={ents} view={w}x{h} variant={variant} passes={passes} first={first} "
                  f"draw={draw} depth={depth} scale={scale} shard={rank}/{world}/{tile} group={on_group} ext={ext} layout={layout} cull={cull} kernel={r.kernel_info()} ndiff={diff.size} first_diff={diff[:4]}", flush=True)
        r.close()
        loader.close()
        if (it + 1) % 2000 == 0:
            print(f"fuzz: {it + 1} of {n_iter} configurations from seed {seed0} so far, {bad} differed", flush=True)
    print(f"fuzz: one-member group ended on {groups[1].transport()['name']}; configurations per transport {transports_used}")
    print(f"fuzz: {n_iter} configurations from seed {seed0}, {bad} differed")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
