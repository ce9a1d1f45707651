#!/bin/bash
# GPU box: the round's evidence — bench line, rocprofv3 kernel stats and PMC summaries of the headline kernel and of the
# entity-BVH kernel, phase profiles, the other configurations.  Everything lands in gpurun_out/r03p/; copy what is to be judged
# into profiles/.
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r03p; mkdir -p $O; cd $R
python3 -c 'from chunkyclplugin_amd import native; native.build(); native.lib(); from oracle import binding; binding.port()' || exit 1
timeout 300 python3 bench.py > $O/bench.json 2> $O/bench.err; tail -1 $O/bench.json | cut -c1-160
timeout 200 python3 tools/phase_stats.py 4 16 > $O/phase_stats_outdoor.json 2>/dev/null
if [ -z "${ONLY_BENCH:-}" ]; then   # ONLY_BENCH=1: the headline kernel's evidence only
CHUNKY_STATS_SCENE=entities timeout 300 python3 tools/phase_stats.py 4 8 > $O/phase_stats_entities.json 2>/dev/null
timeout 1200 python3 tools/config_bench.py benchmark benchmark_entities indoor indoor_nee entities entities4k entities1m > $O/config_bench.jsonl 2> $O/config_bench.err; cut -c1-200 $O/config_bench.jsonl
fi
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_bench -- python3 $R/bench.py --no-cpu --no-roofline --steps 5 > $O/kt_bench.log 2>&1
[ -z "${ONLY_BENCH:-}" ] && (cd $R && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_entities -- python3 tools/config_bench.py entities > $O/kt_entities.log 2>&1)
cd $R
find $O/kt_bench -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/bench_kernel_stats.csv
[ -z "${ONLY_BENCH:-}" ] && find $O/kt_entities -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/entities_kernel_stats.csv
head -4 $O/bench_kernel_stats.csv; [ -z "${ONLY_BENCH:-}" ] && head -4 $O/entities_kernel_stats.csv
PMC_TIMEOUT=200 PMC_GROUPS="sq1 sq2 tcc tcp fetch write" bash tools/pmc.sh r03p_bench > $O/pmc_bench.log 2>&1
cp gpurun_out/pmc_r03p_bench/summary.json $O/pmc_bench_summary.json
if [ -z "${ONLY_BENCH:-}" ]; then
PMC_TIMEOUT=300 PMC_SCRIPT="tools/config_bench.py entities" PMC_GROUPS="sq1 sq2 tcc tcp fetch write" bash tools/pmc.sh r03p_entities > $O/pmc_entities.log 2>&1
cp gpurun_out/pmc_r03p_entities/summary.json $O/pmc_entities_summary.json
fi
rm -rf $O/kt_bench $O/kt_entities
ls $O
