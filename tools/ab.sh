#!/bin/bash
# tools/ab.sh "<variants...>" [what...] — GPU box: A/B of tuning builds made by tools/variants.sh (.variants/; "default" = the in-tree
# library).  what: bench (the headline, 8 steps), eighth (rank 0's share of an 8-GPU split on one GPU), or any tools/config_bench.py
# name (entities, entities1m, indoor, benchmark ...).  An environment override can be tried the same way: "ENV:NAME=value" as a
# variant runs the in-tree library with that variable set (e.g. "ENV:CHUNKY_BVH_LAYOUT=4096,32").
variants=$1; shift
what=${@:-bench}
export CHUNKY_ORACLE_NO_BUILD=1
for v in $variants; do
  unset CHUNKY_HIP_LIB; envset=""
  case "$v" in
    default) ;;
    ENV:*) envset="${v#ENV:}"; export "$envset"; export CHUNKY_HIP_LIB=$PWD/chunkyclplugin_amd/libchunky_hip_tuning.so ;;  # (tuning variables are read by the -DCHUNKY_TUNING build only)
    *) export CHUNKY_HIP_LIB=$PWD/.variants/libchunky_hip_$v.so ;;
  esac
  for w in $what; do
    case "$w" in
      bench)  for rep in 1 2; do timeout 120 python bench.py --no-cpu --no-extras --steps 8 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('bench $v', d['value'], 'launch_ms', d['roofline']['launch_ms'], flush=True)"; done ;;
      eighth) timeout 120 python bench.py --no-cpu --no-extras --steps 8 --emulate-world 8 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('eighth $v', d['value'], 'launch_ms', d['roofline']['launch_ms'], flush=True)" ;;
      *)      timeout 600 python tools/config_bench.py $w 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$w $v', round(d['Msamples/s'],1), 'launch_ms', round(d['launch_ms'],2), 'bit-identical', d['rows_bit_identical_to_oracle'], flush=True)" ;;
    esac
  done
  [ -n "$envset" ] && unset "${envset%%=*}"
done
unset CHUNKY_HIP_LIB
