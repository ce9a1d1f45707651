export CHUNKY_ORACLE_NO_BUILD=1
timeout 900 python bench.py --config 5 --detail gpurun_out/r06_bench_config5_detail.json > gpurun_out/r06_bench_config5.json 2> gpurun_out/r06_bench_config5.err
tail -c 2500 gpurun_out/r06_bench_config5.json; tail -3 gpurun_out/r06_bench_config5.err | cut -c1-300
PMC_TIMEOUT=400 bash tools/pmc.sh bigworld --config 5 > /dev/null 2>&1
python3 -c "
import json; d=json.load(open('gpurun_out/pmc_bigworld/summary.json')); print({k:round(v['mean_per_launch']/1e6,1) for k,v in d['counters'].items()}); print(d['derived'])"
