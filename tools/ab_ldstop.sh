#!/bin/bash
# tools/ab_ldstop.sh <out dir under gpurun_out> — experiment 5.2: entity walk with the BVH tops in LDS (variant 512) against the default
OUT=gpurun_out/$1; mkdir -p $OUT
show() { python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$1', d['config'][:40], 'Msamples/s %.1f' % d['Msamples/s'], 'launch_ms %.2f' % d['launch_ms'], 'identical', d['rows_bit_identical_to_oracle'], d['kernel'])
"; }
for rep in 1; do
  timeout 300 python tools/config_bench.py entities 2>/dev/null | show base | tee -a $OUT/ab.txt
  for tops in ${LDS_TOPS:-320,256 448,127 128,128 1,1}; do
    CHUNKY_BVH_LDS_TOP=$tops CHUNKY_BENCH_KERNEL=512 timeout 300 python tools/config_bench.py entities 2>/dev/null | show "top=$tops" | tee -a $OUT/ab.txt
  done
  CHUNKY_BVH_LDS_TOP=320,256 timeout 300 python tools/config_bench.py entities 2>/dev/null | show "layout-only" | tee -a $OUT/ab.txt
done
