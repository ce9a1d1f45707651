#!/usr/bin/env python3
"""tools/filter_bench.py — device time of the tone-map kernel (`filter`) on HBM-resident buffers.

28 algorithmic bytes per pixel (3 doubles in, one ARGB word out; tonemap/include/post_processing_filter.cl:5-51),
HBM-bound.  Prints one JSON line per (resolution, filter type)."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from chunkyclplugin_amd.renderer import HipPostProcessingFilter, RendererInstance
    inst = RendererInstance.get(0)
    for (w, h) in ((1920, 1080), (3840, 2160), (7680, 4320)):
        n = w * h
        d_in = torch.rand(3 * n, dtype=torch.float64, device="cuda") * 2
        d_out = torch.zeros(n, dtype=torch.int32, device="cuda")
        for fid in ("GAMMA", "TONEMAP1", "TONEMAP2", "TONEMAP3"):
            f = HipPostProcessingFilter(fid, inst)
            f.process_device(n, 1.0, d_in.data_ptr(), d_out.data_ptr(), repeat=3)
            ms = f.process_device(n, 1.0, d_in.data_ptr(), d_out.data_ptr(), repeat=20)
            gbs = 28.0 * n / (ms * 1e-3) / 1e9
            print(json.dumps({"kernel": "filter", "type": fid, "pixels": n, "kernel_ms": round(ms, 5),
                              "achieved_GBs": round(gbs, 1), "frac_of_8TBs": round(gbs / 8000.0, 4)}), flush=True)


if __name__ == "__main__":
    main()
