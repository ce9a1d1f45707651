#!/usr/bin/env python3
"""How far do two CONFORMING OpenCL platforms drift apart on whole images?  (build container only: needs /root/reference)

north_star asks for "per-pixel radiance within 1e-5 relative" of the reference OpenCL kernel.  The reference kernel's
output depends on the platform's sin / cos / asin / acos / atan2 / dot / cross / normalize, which OpenCL only bounds in
ULPs.  This tool links the SAME reference object (rayTracer.cl compiled in place) against two platform layers —
rt_math.h (oracle/_ref/libchunky_ref.so: what the HIP kernels reproduce bit for bit) and glibc libm with unfused vector
builtins (oracle/_ref/libchunky_ref_libm.so, `make -C oracle ref_libm`) — renders the golden scenes at 64 spp with both and
reports, per scene: the fraction of pixels within 1e-5 relative, the worst pixels, and how many of the per-trace hit
records (block indices: integer-exact in north_star) differ.  Result: profiles/r02_tolerance_study.json."""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_scenes as gs  # noqa: E402
from chunkyclplugin_amd import scenes  # noqa: E402
from oracle import binding  # noqa: E402

SPP = 64


def main():
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "ref", "ref_libm"], check=True, stdout=subprocess.DEVNULL)
    ours = binding.RefLib(os.path.join(ROOT, "oracle", "_ref", "libchunky_ref.so"))
    libm = binding.RefLib(os.path.join(ROOT, "oracle", "_ref", "libchunky_ref_libm.so"))
    seeds = scenes.java_random_ints(SPP)
    out = {"spp": SPP, "threshold_rel": 1e-5, "platform_a": "rt_math.h (fused dot/cross, Cephes-style polynomials)",
           "platform_b": "glibc libm, unfused dot / cross / normalize", "scenes": {}}
    tot_px = tot_ok = 0
    for name in gs.NAMES:
        sc = gs.make(name).with_view(128, 96)
        h = binding.SceneHandle(sc)
        a = ours.render_passes(h, seeds).reshape(-1, 3).astype(np.float64)
        b = libm.render_passes(h, seeds).reshape(-1, 3).astype(np.float64)
        rel = np.abs(a - b) / np.maximum(np.abs(a), 1e-6)
        px = rel.max(axis=1)
        ok = px <= 1e-5
        # one pass: pixels whose FIRST sample already differs by more than rounding noise = a flipped decision somewhere on the path
        a1 = ours.render_passes(h, seeds[:1]).reshape(-1, 3).astype(np.float64)
        b1 = libm.render_passes(h, seeds[:1]).reshape(-1, 3).astype(np.float64)
        flip = (np.abs(a1 - b1) / np.maximum(np.abs(a1), 1e-6)).max(axis=1) > 1e-3
        # hit records on a pixel sample: integer fields
        gids = np.arange(0, sc.width * sc.height, 53)
        n_rec = n_diff = 0
        for g in gids:
            ra, _ = ours.trace_records(h, int(seeds[0]), int(g))
            rb, _ = libm.trace_records(h, int(seeds[0]), int(g))
            n_rec += max(len(ra), len(rb))
            if len(ra) != len(rb):
                n_diff += abs(len(ra) - len(rb)) + int((ra["material"][:min(len(ra), len(rb))] != rb["material"][:min(len(ra), len(rb))]).sum())
            else:
                n_diff += int(((ra["material"] != rb["material"]) | (ra["hit"] != rb["hit"])).sum())
        out["scenes"][name] = {"pixels": int(px.size), "within_1e-5": float(ok.mean()), "median_rel": float(np.median(px)),
                               "p99_rel": float(np.quantile(px, 0.99)), "max_rel": float(px.max()),
                               "bit_identical_pixels": float((a == b).all(axis=1).mean()),
                               "first_pass_pixels_with_a_flipped_decision": float(flip.mean()),
                               "trace_records_compared": int(n_rec), "trace_records_with_different_hit_or_block": int(n_diff)}
        tot_px += px.size
        tot_ok += int(ok.sum())
        print(name, out["scenes"][name], flush=True)
    out["all_scenes_within_1e-5"] = tot_ok / tot_px
    out["reading"] = ("Two conforming platforms agree to 1e-5 on most pixels but not all: a last-bit difference in sin/cos/normalize "
                      "moves a bounce direction by an ULP, and every so often that flips a hit decision (another block, another "
                      "texel, sky instead of ground) — a different, equally valid sample.  Hence the parity contract of this repo: "
                      "ONE definition of the platform layer (rt_math.h) shared by the reference build, the C restatement and the "
                      "HIP kernels, and bit equality against it; against a foreign OpenCL driver only the statistics above can hold.")
    path = os.path.join(ROOT, "profiles", "r02_tolerance_study.json")
    json.dump(out, open(path, "w"), indent=1)
    print("written", path, "overall", out["all_scenes_within_1e-5"])


if __name__ == "__main__":
    main()
