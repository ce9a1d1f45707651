#!/bin/bash
# tools/profiles.sh <tag> — GPU box: a round's evidence.  The driver's bench line (stdout) and full result object (--detail) for every
# BASELINE configuration and the beyond-cache world (--config 5), rocprofv3
# kernel stats and PMC summaries (separate --pmc passes: tools/pmc.sh) of the headline kernel and of the entity-BVH kernel, phase
# profiles, the other scenes of tools/config_bench.py.  Everything lands in gpurun_out/<tag>/; copy what is to be judged into
# profiles/ (tools/collect_profiles.py <tag> does that and rebuilds profiles/pmc_traffic.json).   ONLY_BENCH=1: headline only.
TAG=${1:-prof}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
python3 -c 'from chunkyclplugin_amd import native; native.build(); native.lib(); from oracle import binding; binding.port()' || exit 1
export CHUNKY_ORACLE_NO_BUILD=1   # everything is built: nothing below compiles or spawns make (profiled processes must not)
timeout 400 python3 bench.py --detail $O/bench_detail.json > $O/bench.json 2> $O/bench.err; tail -1 $O/bench.json | cut -c1-200
timeout 200 python3 tools/phase_stats.py 4 16 > $O/phase_stats_outdoor.json 2>/dev/null
if [ -z "${ONLY_BENCH:-}" ]; then
for c in 1 3 4 5; do timeout 600 python3 bench.py --config $c --detail $O/bench_config${c}_detail.json > $O/bench_config$c.json 2> $O/bench_config$c.err; tail -1 $O/bench_config$c.json | cut -c1-200; done
CHUNKY_STATS_SCENE=entities timeout 300 python3 tools/phase_stats.py 4 8 > $O/phase_stats_entities.json 2>/dev/null
timeout 1500 python3 tools/config_bench.py benchmark benchmark_entities indoor indoor_nee entities entities4k entities1m > $O/config_bench.jsonl 2> $O/config_bench.err; cut -c1-200 $O/config_bench.jsonl
fi
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_bench -- python3 $R/bench.py --no-cpu --no-roofline --no-extras --steps 5 > $O/kt_bench.log 2>&1
[ -z "${ONLY_BENCH:-}" ] && (cd $R && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_entities -- python3 bench.py --config 4 --no-cpu --no-roofline --no-extras --steps 3 > $O/kt_entities.log 2>&1)
cd $R
find $O/kt_bench -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/bench_kernel_stats.csv
[ -z "${ONLY_BENCH:-}" ] && find $O/kt_entities -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/entities_kernel_stats.csv
head -4 $O/bench_kernel_stats.csv; [ -z "${ONLY_BENCH:-}" ] && head -4 $O/entities_kernel_stats.csv
PMC_TIMEOUT=200 PMC_GROUPS="sq1 sq2 tcc tcp fetch write" bash tools/pmc.sh ${TAG}_bench > $O/pmc_bench.log 2>&1
cp gpurun_out/pmc_${TAG}_bench/summary.json $O/pmc_bench_summary.json
if [ -z "${ONLY_BENCH:-}" ]; then
for c in 1 3 4 5; do
  PMC_TIMEOUT=400 PMC_GROUPS="sq1 sq2 tcc tcp fetch write" bash tools/pmc.sh ${TAG}_config$c --config $c > $O/pmc_config$c.log 2>&1
  cp gpurun_out/pmc_${TAG}_config$c/summary.json $O/pmc_config${c}_summary.json
done
fi
rm -rf $O/kt_bench $O/kt_entities
ls $O
