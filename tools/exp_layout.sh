#!/bin/bash
# GPU box: the entity walk under different placements of the BVH records (CHUNKY_BVH_LAYOUT="top,treelet"; "0,0" = depth-first).
#   tools/exp_layout.sh "0,0 4096,32 ..." [config_bench names...]
layouts=$1; shift
names=${@:-entities}
export CHUNKY_ORACLE_NO_BUILD=1
for l in $layouts; do
  for n in $names; do
    CHUNKY_BVH_LAYOUT=$l timeout 600 python tools/config_bench.py $n 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$n layout $l', round(d['Msamples/s'],1), 'launch_ms', round(d['launch_ms'],2), 'bit-identical', d['rows_bit_identical_to_oracle'], flush=True)"
  done
done
