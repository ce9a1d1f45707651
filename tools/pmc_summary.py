#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs written by tools/pmc.sh: per-launch means for the render kernel.
FETCH_SIZE / WRITE_SIZE are reported in KiB by rocprofv3; HBM bytes follow MI355X_MICROARCH.md section HBM:
read bytes = FETCH_SIZE * 1024 * 2 (gfx950 tallies 128-B requests at 64 B), write bytes = WRITE_SIZE * 1024
(uncalibrated)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root = sys.argv[1]
kern = sys.argv[2] if len(sys.argv) > 2 else "render"
acc = defaultdict(lambda: defaultdict(float))   # counter -> dispatch -> value
names = {}
for path in glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True):
    group = os.path.relpath(path, root).split(os.sep)[0]
    with open(path) as f:
        for row in csv.DictReader(f):
            if kern not in row["Kernel_Name"]:
                continue
            acc[row["Counter_Name"]][(group, row["Dispatch_Id"])] += float(row["Counter_Value"])
            names[row["Counter_Name"]] = row["Kernel_Name"][:60]
out = {}
for c, d in acc.items():
    vals = list(d.values())
    out[c] = {"mean_per_launch": sum(vals) / len(vals), "launches": len(vals)}
m = lambda k: out[k]["mean_per_launch"] if k in out else None
der = {}
if m("SQ_THREAD_CYCLES_VALU") and m("SQ_ACTIVE_INST_VALU"):
    der["valu_lane_utilisation"] = m("SQ_THREAD_CYCLES_VALU") / (m("SQ_ACTIVE_INST_VALU") * 64)
if m("SQ_WAVE_CYCLES"):
    for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_ANY"):
        if m(k):
            der[k + "/WAVE_CYCLES"] = m(k) / m("SQ_WAVE_CYCLES")
if m("TCC_HIT_sum") is not None and m("TCC_MISS_sum") is not None:
    der["l2_hit_rate"] = m("TCC_HIT_sum") / max(m("TCC_HIT_sum") + m("TCC_MISS_sum"), 1)
if m("TCP_TOTAL_CACHE_ACCESSES_sum") and m("TCP_TCC_READ_REQ_sum"):
    der["l1_hit_rate_est"] = 1 - m("TCP_TCC_READ_REQ_sum") / m("TCP_TOTAL_CACHE_ACCESSES_sum")
if m("FETCH_SIZE") is not None:
    der["hbm_read_bytes_per_launch"] = m("FETCH_SIZE") * 1024 * 2
if m("WRITE_SIZE") is not None:
    der["hbm_write_bytes_per_launch"] = m("WRITE_SIZE") * 1024
if "hbm_read_bytes_per_launch" in der and "hbm_write_bytes_per_launch" in der:
    der["hbm_bytes_per_launch"] = der["hbm_read_bytes_per_launch"] + der["hbm_write_bytes_per_launch"]
print(json.dumps({"kernel_filter": kern, "counters": out, "derived": der}, indent=1))
