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
# what actually limits the kernel (the algorithmic roofline above is a work-rate convention: the scene is cache-resident)
tcp_cycles = cycles * 256.0              # one L1 (TCP) per CU
issue = c["SQ_INSTS_VALU"] * 2 / (cycles * 1024)
lim = {
    "valu_issue_frac": round(issue, 4),
    "valu_lane_util": round(d["valu_lane_utilisation"], 4),
    "valu_lane_frac": round(issue * d["valu_lane_utilisation"], 4),   # useful lane-cycles / all VALU lane-cycles of the launch
    "wave_wait_frac": round(d["SQ_WAIT_ANY/WAVE_CYCLES"], 3),
    "l1_tag_lookups_per_cycle": round(c["TCP_TOTAL_CACHE_ACCESSES_sum"] / tcp_cycles, 3) if "TCP_TOTAL_CACHE_ACCESSES_sum" in c else None,
    "l1_pending_stall_frac": round(c["TCP_PENDING_STALL_CYCLES_sum"] / tcp_cycles, 3) if "TCP_PENDING_STALL_CYCLES_sum" in c else None,
    "l2_request_GBps": round(c["TCC_REQ_sum"] * 64 / (cycles / 2.4e9) / 1e9, 1),
    "l2_request_frac_of_34.5TBps": round(c["TCC_REQ_sum"] * 64 / (cycles / 2.4e9) / 34.5e12, 3),
    "l2_hit_rate": round(d["l2_hit_rate"], 4),
    "limiter": "issue + latency: VALU issue slots and the L1's tag look-ups (about one per cycle at most) are both more than half used "
               "while a third of the wave cycles are waits on dependent reads; HBM is at ~3 % of peak",
    "note": "l1_tag_lookups_per_cycle = TCP_TOTAL_CACHE_ACCESSES / (cycles x 256 L1s); l2_request_GBps counts TCC_REQ x 64 B against the "
            "34.5 TB/s aggregate L2 figure of MI355X_MICROARCH.md; cycles = GRBM_GUI_ACTIVE / 8 XCDs at a nominal 2.4 GHz",
}
out["limits"] = lim
print(json.dumps(out, indent=1))
