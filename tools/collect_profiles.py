#!/usr/bin/env python3
"""tools/collect_profiles.py <tag> <prefix> — after tools/profiles.sh <tag> ran on the GPU box: copy gpurun_out/<tag>/'s
summaries to profiles/<prefix>_* (tracked) and rebuild profiles/pmc_traffic.json from the PMC summaries and bench lines."""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, prefix = sys.argv[1], sys.argv[2]
src = os.path.join(ROOT, "gpurun_out", tag)
keep = {"bench.json": "bench.json", "bench_detail.json": "bench_detail.json", "bench_kernel_stats.csv": "render_pool_kernel_stats.csv", "entities_kernel_stats.csv": "entities_kernel_stats.csv",
        "pmc_bench_summary.json": "render_pool_pmc_summary.json", "phase_stats_outdoor.json": "phase_stats_outdoor.json",
        "phase_stats_entities.json": "phase_stats_entities.json", "config_bench.jsonl": "config_bench.jsonl"}
for c in (1, 3, 4, 5):
    keep[f"bench_config{c}.json"] = f"bench_config{c}.json"
    keep[f"bench_config{c}_detail.json"] = f"bench_config{c}_detail.json"
    keep[f"pmc_config{c}_summary.json"] = f"config{c}_pmc_summary.json"
for a, b in keep.items():
    p = os.path.join(src, a)
    if os.path.exists(p) and os.path.getsize(p) > 0:
        shutil.copy(p, os.path.join(ROOT, "profiles", f"{prefix}_{b}"))
        print("profiles/" + f"{prefix}_{b}")
# (the full result objects: the stdout line is a summary since round 6)
for pm, bench in [("render_pool_pmc_summary.json", "bench_detail.json")] + [(f"config{c}_pmc_summary.json", f"bench_config{c}_detail.json") for c in (1, 3, 4, 5)]:
    a, b = os.path.join(ROOT, "profiles", f"{prefix}_{pm}"), os.path.join(ROOT, "profiles", f"{prefix}_{bench}")
    if os.path.exists(a) and os.path.exists(b):
        label = (f"profiles/{prefix}_{pm} (tools/pmc.sh: rocprofv3 --pmc, one pass per counter group, FETCH_SIZE x2 per MI355X_MICROARCH.md section HBM; "
                 "counters of render_pool only — fold_kernel adds its read of the staging array)")
        subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_pmc_traffic.py"), a, b, label], check=False)
