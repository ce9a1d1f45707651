#!/usr/bin/env python3
"""tools/cull_probe.py [traces] [workers] — CHUNKY_OPT_BVH_CULL_BEHIND probed where it CAN differ from the reference's walk
(EXPERIMENTS.md 4.4): a seeded world BVH of axis-aligned entity boxes on the block grid (integer corners: every face lies in a
plane rays leave from), and BVH traces (K/bvh.h:47-109 through oracle/port.c, helper 15) whose origins sit ON those planes — on a
face, exactly at its edge, a few ulps or 1e-7 ... 1e-4 beside it — with grazing directions (normal component 1e-7 ... 1e-1, either
sign, or exactly 0).  Every trace runs with and without port_set_bvh_cull(1); the twelve output words are compared bit for bit.

Prints one JSON line: traces, differing traces, and the first differing rows (input + both outputs).  CPU only."""
import json
import multiprocessing as mp
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

CHUNK = 1 << 18


def box_scene():
    import golden_scenes as gs
    return gs.grid_boxes()


def make_rays(rng, n, lo, hi):
    """n rows of helper 15: origin, direction, limit — origins on face planes of the boxes, grazing directions."""
    b = rng.integers(0, len(lo), n)
    axis = rng.integers(0, 3, n)
    side = rng.integers(0, 2, n)
    plane = np.where(side == 1, hi[b, axis], lo[b, axis]).astype(np.float32)
    o = np.zeros((n, 3), np.float32)
    d = np.zeros((n, 3), np.float64)
    # tangential coordinates: inside the face / exactly on an edge / a hair beside an edge (either side)
    for k in range(3):
        l, h = lo[b, k], hi[b, k]
        inside = (l + rng.random(n) * (h - l)).astype(np.float32)
        edge = np.where(rng.random(n) < 0.5, l, h).astype(np.float32)
        hair = (edge.astype(np.float64) + rng.choice([-1.0, 1.0], n) * 10.0 ** rng.uniform(-7, -4, n)).astype(np.float32)
        ulps = edge.copy()
        for _ in range(3):
            step = rng.random(n) < 0.6
            ulps = np.where(step, np.nextafter(ulps, np.where(rng.random(n) < 0.5, np.float32(-1e9), np.float32(1e9)).astype(np.float32)), ulps)
        kind = rng.integers(0, 6, n)
        o[:, k] = np.where(kind < 3, inside, np.where(kind == 3, edge, np.where(kind == 4, hair, ulps)))
    # the normal coordinate: exactly on the plane, a few ulps off, or where point = o + d (t - OFFSET) leaves it (1e-7 ... 1e-4 off)
    rows = np.arange(n)
    pk = rng.integers(0, 4, n)
    off = plane.copy()
    for _ in range(4):
        step = (pk == 1) & (rng.random(n) < 0.6)
        off = np.where(step, np.nextafter(off, np.where(rng.random(n) < 0.5, np.float32(-1e9), np.float32(1e9)).astype(np.float32)), off)
    far = (plane.astype(np.float64) + rng.choice([-1.0, 1.0], n) * 10.0 ** rng.uniform(-7, -4, n)).astype(np.float32)
    o[rows, axis] = np.where(pk == 0, plane, np.where(pk == 1, off, far))
    # direction: grazing — the normal component tiny (or exactly zero), the rest a random unit vector in the plane
    t = rng.normal(size=(n, 3))
    t[rows, axis] = 0
    t /= np.linalg.norm(t, axis=1, keepdims=True)
    dn = rng.choice([-1.0, 1.0], n) * 10.0 ** rng.uniform(-7, -1, n)
    dn = np.where(rng.random(n) < 0.1, 0.0, dn)
    d[:] = t * np.sqrt(np.maximum(1 - dn * dn, 0))[:, None]
    d[rows, axis] = dn
    lim = np.where(rng.random(n) < 0.5, np.float32(np.inf), rng.uniform(0.01, 6.0, n).astype(np.float32))
    out = np.zeros((n, 32), np.float32)
    out[:, 0:3] = o
    out[:, 3:6] = d.astype(np.float32)
    out[:, 6] = lim
    return out


def worker(args):
    seed, n = args
    from oracle import binding
    import ctypes as C
    port = binding.port()
    port.lib.port_set_bvh_cull.argtypes = [C.c_int]
    port.lib.port_set_bvh_cull.restype = None
    sc, lo, hi = box_scene()
    h = binding.SceneHandle(sc)
    rng = np.random.default_rng(seed)
    done = differ = hits = 0
    first = []
    while done < n:
        m = min(CHUNK, n - done)
        rays = make_rays(rng, m, lo, hi)
        port.lib.port_set_bvh_cull(0)
        a = port.helpers(h, 15, rays)
        port.lib.port_set_bvh_cull(1)
        c = port.helpers(h, 15, rays)
        port.lib.port_set_bvh_cull(0)
        bad = (a.view(np.uint32) != c.view(np.uint32)).any(axis=1)
        differ += int(bad.sum())
        hits += int((a[:, 0] != 0).sum())
        for i in np.flatnonzero(bad)[:3]:
            if len(first) < 3:
                first.append({"row": rays[i, :7].view(np.uint32).tolist(), "row_f": [float(x) for x in rays[i, :7]], "reference": [float(x) for x in a[i]], "culled": [float(x) for x in c[i]]})
        done += m
    return done, differ, hits, first


def main():
    n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10 ** 8
    workers = int(sys.argv[2]) if len(sys.argv) > 2 else max(1, len(os.sched_getaffinity(0)))
    per = (n + workers - 1) // workers
    with mp.get_context("fork").Pool(workers) as pool:
        res = pool.map(worker, [(1000 + w, per) for w in range(workers)])
    import golden_scenes as gs
    out = {"scene": f"{gs.GRID_BOXES} grid-aligned boxes ({12 * gs.GRID_BOXES} triangles) in the world BVH (tests/golden_scenes.py grid_boxes)", "traces": sum(r[0] for r in res),
           "differing": sum(r[1] for r in res), "traces_that_hit": sum(r[2] for r in res), "first_differing": [x for r in res for x in r[3]][:6]}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
