#!/bin/bash
# GPU box: rank 0's share of an N-GPU tile split rendered on one GPU (bench.py --emulate-world N), N = 1 2 4 8:
# kernel time per step should fall as 1/N if a share keeps the GPU as busy as the whole image does.
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r02s; mkdir -p $O; cd $R
for n in 1 2 4 8; do
  if [ $n = 1 ]; then E=""; else E="--emulate-world $n"; fi
  timeout 200 python3 bench.py --no-cpu --no-roofline --steps 4 $E 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); print(json.dumps({'world': $n, 'Msamples/s_of_share': d['value'], 'ms_per_step': d['ms_per_step'], 'launch_ms': d['roofline']['launch_ms'], 'kernel': d['roofline']['kernel']}))" | tee -a $O/emulated_scaling.jsonl
done
