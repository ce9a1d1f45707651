#!/usr/bin/env python3
"""tools/make_pmc_traffic.py <pmc summary json> <bench json line file> <source label> > profiles/pmc_traffic.json

Builds the file bench.py reads for `roofline.traffic` / `roofline.valu` from a PMC summary of tools/pmc.sh and the bench line
of the same build: the launch shape (kernel, passes and samples per launch) is recorded so that bench.py attaches the
counters only to runs of that shape."""
import json
import sys

pm = json.load(open(sys.argv[1]))
bench = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
c = {k: v["mean_per_launch"] for k, v in pm["counters"].items()}
d = pm["derived"]
roof = bench["roofline"]
name = roof["kernel"]                     # e.g. render_pool<17,56>+fold_kernel
tree, pool = [int(x) for x in name[name.index("<") + 1:name.index(">")].split(",")[:2]]
cycles = c["GRBM_GUI_ACTIVE"] / 8.0      # summed over the 8 XCDs
out = {
    "kernel": f"{name}, {bench['config']['passes_per_step']} passes per launch, 1920x1080, 1 rank",
    "kernel_info": [tree, 1, 0, pool],
    "passes_per_launch": bench["config"]["passes_per_step"],
    "samples_per_launch": roof["samples_per_launch"],
    "source": sys.argv[3],
    "hbm_read_bytes_per_launch": d["hbm_read_bytes_per_launch"],
    "hbm_write_bytes_per_launch": d["hbm_write_bytes_per_launch"],
    "hbm_bytes_per_launch": int(d["hbm_bytes_per_launch"]),
    "valu": {
        "issue_frac": round(c["SQ_INSTS_VALU"] * 2 / (cycles * 1024), 4),
        "lane_util": round(d["valu_lane_utilisation"], 4),
        "wait_frac_of_wave_cycles": round(d["SQ_WAIT_ANY/WAVE_CYCLES"], 3),
        "salu_per_valu": round(c["SQ_INSTS_SALU"] / c["SQ_INSTS_VALU"], 3),
        "l2_hit_rate": round(d["l2_hit_rate"], 4),
        "l1_hit_rate_est": round(d.get("l1_hit_rate_est", 0.0), 4),
        "note": "issue_frac = SQ_INSTS_VALU x 2 cycles (wave64 on SIMD-32) / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs); "
                "lane_util = SQ_THREAD_CYCLES_VALU / (SQ_ACTIVE_INST_VALU x 64)",
    },
}
print(json.dumps(out, indent=1))
