export CHUNKY_ORACLE_NO_BUILD=1
bash tools/ab.sh "base form0 default form3" bench benchmark indoor > gpurun_out/r06_staged_ab2.txt 2>&1
for v in base form0 default; do
  if [ $v != default ]; then export CHUNKY_HIP_LIB=$PWD/.variants/libchunky_hip_$v.so; else unset CHUNKY_HIP_LIB; fi
  PMC_GROUPS="sq1 sq2" PMC_SCRIPT="tools/config_bench.py benchmark" bash tools/pmc.sh city_$v > /dev/null 2>&1
done
unset CHUNKY_HIP_LIB
cat gpurun_out/r06_staged_ab2.txt
python3 - <<'P'
import json
for v in ("base","form0","default"):
    d=json.load(open(f"gpurun_out/pmc_city_{v}/summary.json"))
    c={k:round(x["mean_per_launch"]/1e6,1) for k,x in d["counters"].items()}
    print(v, c, {k:round(x,3) for k,x in d["derived"].items()})
P
