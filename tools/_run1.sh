export CHUNKY_ORACLE_NO_BUILD=1
for v in nosplit default s64w7; do
  if [ $v != default ]; then export CHUNKY_HIP_LIB=$PWD/.variants/libchunky_hip_$v.so; else unset CHUNKY_HIP_LIB; fi
  for sc in outdoor city; do
    CHUNKY_STATS_SCENE=$sc timeout 200 python tools/phase_stats.py 4 > gpurun_out/r06_split_stats_${v}_$sc.json 2>/dev/null
    python3 - $v $sc <<'P'
import json,sys
v,sc=sys.argv[1:3]
d=json.load(open(f"gpurun_out/r06_split_stats_{v}_{sc}.json"))
print(v, sc, "Ms/s %.0f"%d["Msamples/s"], {k:(round(d[k]["lanes_per_exec"],1), round(d[k]["cycles_per_exec"]), round(d[k]["time_share"],3)) for k in ("march","block","shade")}, "swap", d["parts_share_of_total"]["swap"], "swapped/sample", round(d["swaps"]["paths_swapped_per_sample"],2), d["parts_share_of_total"])
P
  done
done
unset CHUNKY_HIP_LIB
bash tools/ab.sh "snt snt64w7" bench benchmark indoor
PMC_GROUPS="tcc tcp" bash tools/pmc.sh split_default > /dev/null 2>&1
CHUNKY_HIP_LIB=$PWD/.variants/libchunky_hip_nosplit.so PMC_GROUPS="tcc tcp" bash tools/pmc.sh split_nosplit > /dev/null 2>&1
python3 - <<'P'
import json
for v in ("nosplit","default"):
    d=json.load(open(f"gpurun_out/pmc_split_{v}/summary.json"))
    print(v, {k:round(x["mean_per_launch"]/1e6,1) for k,x in d["counters"].items()}, {k:round(x,3) for k,x in d["derived"].items()})
P
