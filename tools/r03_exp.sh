#!/bin/bash
# GPU box: A/B of tuning builds (.variants/) — bench.py for the outdoor kernel, tools/config_bench.py entities for the walk.
#   tools/r03_exp.sh "bench variants..." "entity variants..."     ("default" = the in-tree library)
for v in $1; do
  if [ "$v" = default ]; then unset CHUNKY_HIP_LIB; else export CHUNKY_HIP_LIB=$PWD/.variants/libchunky_hip_$v.so; fi
  for rep in 1 2; do
    timeout 90 python bench.py --no-cpu --steps 8 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('bench $v', d['value'], 'launch_ms', d['roofline']['launch_ms'])"
  done
done
for v in $2; do
  if [ "$v" = default ]; then unset CHUNKY_HIP_LIB; else export CHUNKY_HIP_LIB=$PWD/.variants/libchunky_hip_$v.so; fi
  timeout 300 python tools/config_bench.py entities 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('entities $v', round(d['Msamples/s'],1), 'launch_ms', round(d['launch_ms'],2), 'bit-identical', d['rows_bit_identical_to_oracle'])"
done
unset CHUNKY_HIP_LIB
