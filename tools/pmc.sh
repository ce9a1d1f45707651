#!/bin/bash
# tools/pmc.sh <tag> [bench args...] — PMC counter passes for the render kernel, one rocprofv3 run per group
# (SQ has 8 slots, TCC 4; FETCH_SIZE costs 3, WRITE_SIZE 2 — MI355X_MICROARCH.md "rocprofv3 PMC slots").
# Run on the GPU box through gpurun; results land in gpurun_out/pmc_<tag>/<group>/ and a summary in
# gpurun_out/pmc_<tag>/summary.json (tools/pmc_summary.py).
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
# build everything BEFORE the profiler is attached: nothing may compile or spawn a child from inside a profiled process
# (with --pmc the profiler's preloaded library initialises the GPU in every child, and a child that then execs is refused)
(cd $R && python3 -c 'from chunkyclplugin_amd import native; native.build(); native.lib(); from oracle import binding; binding.port()') || exit 1
export CHUNKY_ORACLE_NO_BUILD=1   # the profiled process (and tools/config_bench.py's oracle check in it) never spawns make
cd /tmp && export TMPDIR=/tmp
declare -A G
G[sq1]="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY"
G[sq2]="SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_INST_LEVEL_VMEM SQ_INSTS_BRANCH"
G[tcc]="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE"
G[tcp]="TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_READ_sum TCP_PENDING_STALL_CYCLES_sum"
G[ta]="TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum"
G[tcp2]="TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCP_TA_ADDR_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"
G[tcp3]="TCP_TCP_LATENCY_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TOTAL_ACCESSES_sum"
G[tlb]="TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_TOTAL_CACHE_ACCESSES_sum"
G[td]="TD_TD_BUSY_sum TD_TC_STALL_sum TD_LOAD_WAVEFRONT_sum GRBM_GUI_ACTIVE"
G[fetch]="FETCH_SIZE"
G[write]="WRITE_SIZE"
for g in ${PMC_GROUPS:-sq1 sq2 tcc tcp fetch write}; do
  # PMC_SCRIPT=<python file + args> profiles that instead of the bench (e.g. "tools/config_bench.py entities"); the program
  # after -- is python3 itself (never env / bash -c: the profiler's library initialises the GPU before the program starts)
  if [ -n "${PMC_SCRIPT:-}" ]; then
    (cd $R && timeout ${PMC_TIMEOUT:-240} rocprofv3 --pmc ${G[$g]} --output-format csv -d $OUT/$g -- python3 $PMC_SCRIPT > $OUT/$g.log 2>&1) || echo "group $g failed"
  else
    timeout ${PMC_TIMEOUT:-240} rocprofv3 --pmc ${G[$g]} --output-format csv -d $OUT/$g -- python3 $R/bench.py --no-cpu --no-roofline --no-extras --steps 2 --warmup 1 "$@" > $OUT/$g.log 2>&1 || echo "group $g failed"
  fi
done
python3 $R/tools/pmc_summary.py $OUT ${PMC_KERNEL:-render} > $OUT/summary.json
cat $OUT/summary.json
