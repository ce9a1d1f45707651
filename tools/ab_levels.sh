#!/bin/bash
# tools/ab_levels.sh <config> "<split> <split> ..." — GPU box: level splits of the wide tree (CHUNKY_WIDE_LEVELS="top,l1,..."), all in the
# generic tree form (--kernel 128: render_pool<-1,32>) so that only the split differs.  e.g. tools/ab_levels.sh 2 "6,3 7,2 5,4"
cfg=${1:-2}; shift
for lv in ${1:-"6,3 7,2 5,4"}; do
  for rep in 1 2; do
    CHUNKY_HIP_LIB=$PWD/chunkyclplugin_amd/libchunky_hip_tuning.so CHUNKY_WIDE_LEVELS=$lv timeout 160 python bench.py --config $cfg --no-cpu --no-extras --steps 6 --kernel 128 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('config $cfg levels $lv', d['value'], 'launch_ms', d['roofline']['launch_ms'], d['roofline']['kernel'], flush=True)"
  done
done
