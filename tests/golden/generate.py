"""Regenerates tests/golden/*.npz by running the REFERENCE kernel (oracle/_ref: rayTracer.cl compiled
in place from /root/reference for x86-64).  Only works in the build container.

    python tests/golden/generate.py

Each file holds, for one named scene of tests/golden_scenes.py: the sha256 of the regenerated
inputs, the `res` buffer after N_PASSES passes with the java.util.Random(0) seed stream, the
`preview` ARGB image, and per-trace hit records + radiance for RECORD_GIDS at seed[0].
kats.npz holds the function-level known answers (PCG stream, seed stream, sun basis, builtins).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))

import golden_scenes as gs  # noqa: E402
from chunkyclplugin_amd import scenes  # noqa: E402
from oracle import binding  # noqa: E402


def main():
    ref = binding.ref()
    assert ref is not None, "needs /root/reference"
    seeds = scenes.java_random_ints(gs.N_PASSES)
    for name in gs.NAMES:
        sc = gs.make(name)
        h = binding.SceneHandle(sc)
        res = ref.render_passes(h, seeds)
        prev = ref.preview(h)
        recs = np.zeros((len(gs.RECORD_GIDS), binding.MAX_TRACES), binding.HIT_DTYPE)
        cnt = np.zeros(len(gs.RECORD_GIDS), np.int32)
        rad = np.zeros((len(gs.RECORD_GIDS), 3), np.float32)
        for i, g in enumerate(gs.RECORD_GIDS):
            r, c = ref.trace_records(h, int(seeds[0]), int(g))
            recs[i, :len(r)] = r
            cnt[i] = len(r)
            rad[i] = c
        np.savez_compressed(os.path.join(HERE, name + ".npz"), digest=gs.input_digest(sc), seeds=seeds, res=res,
                            preview=prev, records=recs, counts=cnt, radiance=rad)
        print(name, "res mean", float(np.nanmean(res)), "traces", int(cnt.sum()))
    # function-level known answers
    st, fl = ref.pcg_stream(0, 16)
    st2, fl2 = ref.pcg_stream(0xDEADBEEF, 16)
    xs = np.linspace(-7, 7, 4001).astype(np.float32)
    us = np.linspace(-1.2, 1.2, 4001).astype(np.float32)
    rng = np.random.default_rng(1)
    ya, xa = rng.normal(size=4000).astype(np.float32), rng.normal(size=4000).astype(np.float32)
    sun = scenes.pack_sun(0.174533, 1.256637, 1.25, True)
    np.savez_compressed(
        os.path.join(HERE, "kats.npz"), pcg_states0=st, pcg_floats0=fl, pcg_states1=st2, pcg_floats1=fl2,
        java_seeds=scenes.java_random_ints(8), xs=xs, us=us, ya=ya, xa=xa,
        sin=ref.math(0, xs), cos=ref.math(1, xs), asin=ref.math(2, us), acos=ref.math(3, us),
        atan2=ref.math(4, ya, xa), fmod1=ref.math(5, xs), sun=sun, sun_basis=ref.sun_basis(sun))
    print("kats written")


if __name__ == "__main__":
    main()
