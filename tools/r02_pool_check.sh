#!/bin/bash
# GPU box: first contact of render_pool — small parity first (under a short timeout: a hang must not take the box), then
# everything, then the bench for each pool size and the grouped kernel.
O=gpurun_out/${1:-r02b}; mkdir -p $O
timeout 120 python -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 60 -k "goldens and outdoor-0" 2>&1 | tail -5 > $O/first.log
cat $O/first.log
grep -q passed $O/first.log || exit 1
(timeout 900 python -m pytest tests -m gpu -q -x --timeout 300 2>&1 | tail -8) > $O/pytest.log; cat $O/pytest.log
for v in 0 64 128 192 8; do
  timeout 200 python bench.py --no-cpu --steps 4 --kernel $v 2>$O/bench_$v.err | tail -1 > $O/bench_$v.json
  python - <<PY
import json
d=json.load(open("$O/bench_$v.json")); print("variant $v", d["value"], "Msamples/s launch_ms", d["roofline"]["launch_ms"], d["roofline"]["kernel"])
PY
done
timeout 200 python tools/phase_stats.py 4 16 > $O/phase_stats.json 2> $O/phase_stats.err; cat $O/phase_stats.json
