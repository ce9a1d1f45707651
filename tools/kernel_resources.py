#!/usr/bin/env python3
"""tools/kernel_resources.py [out.json] — registers, spills and occupancy of every render_pool / fold_kernel instantiation as hipcc
reports them for gfx950 (-Rpass-analysis=kernel-resource-usage on csrc/render_pool.hip with the library's flags).  CPU only."""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from chunkyclplugin_amd import native  # noqa: E402

flags = [f for f in native.HIPCC_FLAGS if f not in ("-shared",)]
cmd = ["hipcc", *flags, "-x", "hip", "-c", os.path.join(native.CSRC, "render_pool.hip"), "-o", os.devnull, "-Rpass-analysis=kernel-resource-usage"]
err = subprocess.run(cmd, capture_output=True, text=True).stderr
out, cur = {}, None
for line in err.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = out.setdefault(re.sub(r"\(.*", "", name).replace("chunky::", ""), {})
        continue
    for key, pat in (("sgprs", r"TotalSGPRs: (\d+)"), ("vgprs", r" VGPRs: (\d+)"), ("scratch_bytes_per_lane", r"ScratchSize \[bytes/lane\]: (\d+)"),
                     ("waves_per_simd", r"Occupancy \[waves/SIMD\]: (\d+)"), ("sgpr_spills", r"SGPRs Spill: (\d+)"), ("vgpr_spills", r"VGPRs Spill: (\d+)")):
        m = re.search(pat, line)
        if m and cur is not None:
            cur[key] = int(m.group(1))
res = {"source": "hipcc -Rpass-analysis=kernel-resource-usage, flags " + " ".join(flags), "kernels": out}
path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "kernel_resources.json")
json.dump(res, open(path, "w"), indent=1)
print(path, len(out), "kernels")
