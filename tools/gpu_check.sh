#!/bin/bash
# tools/gpu_check.sh [variant] — GPU parity tests + phase profile + short bench, every step under a timeout.
V=${1:-0}
(timeout 300 python -m pytest tests -m gpu -q -x --timeout 90) 2>&1 | tail -2
timeout 120 python tools/phase_stats.py $((V | 4)) | python -c "
import json,sys; d=json.load(sys.stdin); print('stats launch_ms %.3f Msamples/s %.1f' % (d['launch_ms'], d['Msamples/s'])); [print(k, {a:round(b,2) for a,b in d[k].items()}) for k in ('march','block','shade','handover','waves')]; print('parts', d.get('parts_share_of_total'))"
(timeout 200 python bench.py --no-cpu --no-extras --steps 8 --kernel $V) 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.readline()); print('bench variant $V', 'Msamples/s', d['value'], 'launch_ms', d['roofline']['launch_ms'], 'frac', d['roofline']['frac'])"
