#!/bin/bash
# GPU box: quick parity of the default build, its bench + phase profile, then the tuning builds of tools/variants.sh
O=gpurun_out/${1:-r02d}; shift; mkdir -p $O
(timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_timed_kernels.py -m gpu -q -x --timeout 200 2>&1 | tail -3) > $O/pytest.log; cat $O/pytest.log
timeout 200 python bench.py --no-cpu --steps 5 2>$O/bench.err | tail -1 > $O/bench.json
python -c "
import json; d=json.load(open('$O/bench.json')); print('default', d['value'], d['roofline']['launch_ms'], d['roofline']['kernel'])"
timeout 200 python tools/phase_stats.py 4 16 > $O/phase_stats.json 2> $O/phase_stats.err
python -c "
import json; d=json.load(open('$O/phase_stats.json'))
for k in ('march','block','shade','model','swaps','loop','parts_share_of_total'): print(k, {a:(round(b,3) if isinstance(b,float) else b) for a,b in d.get(k,{}).items()})"
bash tools/sweep.sh "$@" | tee $O/sweep.txt
