#!/bin/bash
# GPU box: A/B of tuning builds on the whole image and on an emulated eighth share (rank 0 of 8 on one GPU)
for v in $1; do
  if [ "$v" = default ]; then unset CHUNKY_HIP_LIB; else export CHUNKY_HIP_LIB=$PWD/.variants/libchunky_hip_$v.so; fi
  timeout 90 python bench.py --no-cpu --steps 8 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('bench $v', d['value'], 'launch_ms', d['roofline']['launch_ms'])"
  timeout 90 python bench.py --no-cpu --steps 8 --emulate-world 8 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('eighth $v', d['value'], 'launch_ms', d['roofline']['launch_ms'])"
done
unset CHUNKY_HIP_LIB
