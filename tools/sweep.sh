#!/bin/bash
# tools/sweep.sh NAME... — short bench of each tuning build made by tools/variants.sh (run on the GPU box)
for v in "$@"; do
  CHUNKY_HIP_LIB=$PWD/.variants/libchunky_hip_$v.so timeout 200 python bench.py --no-cpu --steps 5 2>&1 | tail -1 | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.readline()); print('$v', d['value'], d['roofline']['launch_ms'])
except Exception as e: print('$v', 'failed', e)"
done
