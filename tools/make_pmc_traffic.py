#!/usr/bin/env python3
"""tools/make_pmc_traffic.py <pmc summary json> <bench json line file> <source label> [profiles/pmc_traffic.json]

Adds (or replaces) the entry of one BASELINE configuration in profiles/pmc_traffic.json — the file bench.py reads for
`roofline.traffic` / `roofline.valu` / `roofline.limits` — from a PMC summary of tools/pmc.sh and the bench line of the same
build and configuration.  The launch shape (kernel, passes and samples per launch) is recorded so that bench.py attaches the
counters only to runs of that shape."""
import json
import os
import sys

pm = json.load(open(sys.argv[1]))
_text = open(sys.argv[2]).read().strip()
try:
    bench = json.loads(_text)                      # the full result object (bench_detail.json)
except ValueError:
    bench = json.loads(_text.splitlines()[-1])     # ... or a file whose last line is one
target = sys.argv[4] if len(sys.argv) > 4 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "pmc_traffic.json")
c = {k: v["mean_per_launch"] for k, v in pm["counters"].items()}
d = pm["derived"]
roof = bench["roofline"]
name = roof["kernel"]                     # e.g. render_pool<17,56>+fold_kernel or render_pool<17,16,bvh>+fold_kernel
parts = name[name.index("<") + 1:name.index(">")].split(",")
tree, pool, bvh = int(parts[0]), int(parts[1]), int(len(parts) > 2 and parts[2] == "bvh")
sorted_ = len(parts) > 2 and parts[2] == "sorted"
cycles = c["GRBM_GUI_ACTIVE"] / 8.0      # summed over the 8 XCDs
tcp_cycles = cycles * 256.0              # one L1 (TCP) per CU
issue = c["SQ_INSTS_VALU"] * 2 / (cycles * 1024)
lane = d["valu_lane_utilisation"]
wait = d["SQ_WAIT_ANY/WAVE_CYCLES"]
lookups = c["TCP_TOTAL_CACHE_ACCESSES_sum"] / tcp_cycles if "TCP_TOTAL_CACHE_ACCESSES_sum" in c else None
hbm = d.get("hbm_bytes_per_launch")
launch_s = roof["launch_ms"] * 1e-3
# the L1's look-up rate: tools/ubench_gather.hip measures 0.64 cycles per distinct line of a gather (profiles/r03_ubench_gather.jsonl),
# i.e. at most 1.56 look-ups per cycle and CU — "about one" was too low (configs[3] ran 1.11)
L1_LOOKUPS_PEAK = 1.0 / 0.64
cu_cycles = cycles * 256.0
salu = c["SQ_INSTS_SALU"] / cu_cycles
branch = c.get("SQ_INSTS_BRANCH", 0.0) / cu_cycles
smem = c.get("SQ_INSTS_SMEM", 0.0) / cu_cycles
# registers of the instantiation (tools/kernel_resources.py)
regs = None
try:
    kr = json.load(open(os.path.join(os.path.dirname(target), "kernel_resources.json")))["kernels"]
    key = "void render_pool<%d, %d, false, %s, false, %s>" % (tree, pool, "true" if bvh else "false", "true" if sorted_ else "false")
    regs = kr.get(key)
except Exception:
    pass
import datetime
import subprocess
try:
    commit = subprocess.run(["git", "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True, cwd=os.path.dirname(os.path.abspath(__file__))).stdout.strip()
except Exception:
    commit = None
collected = {"commit": commit, "box": bench.get("box"), "date": datetime.date.today().isoformat(),
             "note": "counters of a profiled run on this box at this commit; a bench line that carries them says so in pmc_source — "
                     "they are not re-measured in the driver's run (a profiled pass costs 6 launches per counter group)"}
if lookups is not None and lookups / L1_LOOKUPS_PEAK > 0.85:
    limiter = ("the L1s' tag look-ups (%.2f per cycle and L1, about one at most: every lane of a scattered load is a look-up of its own); "
               "VALU lanes %.0f %% useful, waves waiting %.0f %% of their cycles, L2 requests at %.0f %% of its bandwidth"
               % (lookups, 100 * issue * lane, 100 * wait, 100 * c["TCC_REQ_sum"] * 64 / (cycles / 2.4e9) / 34.5e12))
else:
    limiter = ("issue + latency: VALU issue slots %.0f %% used (at %.0f %% of the lanes), the scalar pipe %.0f %% (SALU %.2f + branch %.2f per CU-cycle), "
               "the L1s' look-ups %.0f %% of the measured gather rate (%.2f of 1.56 per cycle) while %.0f %% of the wave cycles are waits on dependent "
               "reads; no unit is saturated — throughput follows paths in flight x latency; HBM is at %.1f %% of peak"
               % (100 * issue, 100 * lane, 100 * (salu + branch), salu, branch, 100 * (lookups or 0) / L1_LOOKUPS_PEAK, lookups or 0, 100 * wait,
                  100 * (hbm or 0) / launch_s / 8e12))
entry = {
    "config": bench["config"].get("baseline_config", 2),
    "kernel": f"{name}, {bench['config']['passes_per_step']} passes per launch, {bench['config']['workload']}",
    "kernel_info": [tree, 1, bvh, pool],
    "passes_per_launch": bench["config"]["passes_per_step"],
    "samples_per_launch": roof["samples_per_launch"],
    "source": sys.argv[3],
    "collected": collected,
    "registers": regs,
    "hbm_read_bytes_per_launch": d.get("hbm_read_bytes_per_launch"),
    "hbm_write_bytes_per_launch": d.get("hbm_write_bytes_per_launch"),
    "hbm_bytes_per_launch": int(hbm) if hbm else None,
    "valu": {
        "issue_frac": round(issue, 4), "lane_util": round(lane, 4), "wait_frac_of_wave_cycles": round(wait, 3),
        "salu_per_valu": round(c["SQ_INSTS_SALU"] / c["SQ_INSTS_VALU"], 3), "l2_hit_rate": round(d["l2_hit_rate"], 4),
        "l1_hit_rate_est": round(d.get("l1_hit_rate_est", 0.0), 4),
        "note": "issue_frac = SQ_INSTS_VALU x 2 cycles (wave64 on SIMD-32) / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs); "
                "lane_util = SQ_THREAD_CYCLES_VALU / (SQ_ACTIVE_INST_VALU x 64)",
    },
    # what actually limits the kernel (the algorithmic roofline is a work-rate convention: the scene is cache-resident)
    "limits": {
        "valu_issue_frac": round(issue, 4), "valu_lane_util": round(lane, 4),
        "valu_lane_frac": round(issue * lane, 4),   # useful lane-cycles / all VALU lane-cycles of the launch
        "wave_wait_frac": round(wait, 3),
        "l1_tag_lookups_per_cycle": round(lookups, 3) if lookups is not None else None,
        "l1_tag_lookup_frac": round(lookups / L1_LOOKUPS_PEAK, 3) if lookups is not None else None,
        "salu_per_cu_cycle": round(salu, 3), "branch_per_cu_cycle": round(branch, 3), "smem_per_cu_cycle": round(smem, 4),
        "scalar_pipe_frac": round(salu + branch, 3),
        "sgprs": regs.get("sgprs") if regs else None, "vgprs": regs.get("vgprs") if regs else None,
        "l1_pending_stall_frac": round(c["TCP_PENDING_STALL_CYCLES_sum"] / tcp_cycles, 3) if "TCP_PENDING_STALL_CYCLES_sum" in c else None,
        "l2_request_GBps": round(c["TCC_REQ_sum"] * 64 / (cycles / 2.4e9) / 1e9, 1),
        "l2_request_frac_of_34.5TBps": round(c["TCC_REQ_sum"] * 64 / (cycles / 2.4e9) / 34.5e12, 3),
        "l2_hit_rate": round(d["l2_hit_rate"], 4),
        "l2_miss_bytes_per_launch": int(c["TCC_MISS_sum"] * 128),
        "limiter": limiter,
        "note": "l1_tag_lookups_per_cycle = TCP_TOTAL_CACHE_ACCESSES / (cycles x 256 L1s), l1_tag_lookup_frac = that over the 1.56 per cycle a gather "
                "sustains (tools/ubench_gather.hip); scalar_pipe_frac = (SQ_INSTS_SALU + SQ_INSTS_BRANCH) / CU-cycles (one scalar issue per CU and cycle); "
                "l2_request_GBps counts TCC_REQ x 64 B against the "
                "34.5 TB/s aggregate L2 figure of MI355X_MICROARCH.md; cycles = GRBM_GUI_ACTIVE / 8 XCDs at a nominal 2.4 GHz",
    },
}
entries = []
if os.path.exists(target):
    try:
        old = json.load(open(target))
        entries = old.get("entries", [old] if "kernel_info" in old else [])
    except Exception:
        entries = []
entries = [e for e in entries if e.get("config", 2) != entry["config"]] + [entry]
entries.sort(key=lambda e: e.get("config", 2))
json.dump({"entries": entries}, open(target, "w"), indent=1)
print(f"{target}: {len(entries)} entries; config {entry['config']}: {entry['kernel']}")
