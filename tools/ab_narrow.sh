#!/bin/bash
# tools/ab_narrow.sh <out dir under gpurun_out> — experiment 5.4: the octree's last level in 1- / 2-byte entries (variant 1024)
OUT=gpurun_out/$1; mkdir -p $OUT
line() { python -c "
import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); ic=d.get('image_check') or {}
print('$1', d['roofline']['kernel'], 'Msamples/s', d['value'], 'launch_ms', d['roofline']['launch_ms'], 'identical', ic.get('bit_identical'), ic.get('pixels'))"; }
for rep in 1 2; do
  for c in 2 1 3; do
    timeout 200 python bench.py --config $c --no-cpu --no-roofline --steps 6 --no-other-configs --no-extras 2>/dev/null | line "config $c default" | tee -a $OUT/ab.txt
    CHUNKY_NARROW_TREE=1 timeout 200 python bench.py --config $c --no-cpu --no-roofline --steps 6 --kernel 1024 --no-other-configs 2>/dev/null | line "config $c narrow " | tee -a $OUT/ab.txt
    [ $c != 1 ] && CHUNKY_NARROW_TREE=2 timeout 200 python bench.py --config $c --no-cpu --no-roofline --steps 6 --kernel 1024 --no-other-configs 2>/dev/null | line "config $c narrow2" | tee -a $OUT/ab.txt
  done
done
