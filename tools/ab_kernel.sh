#!/bin/bash
# tools/ab_kernel.sh <out dir under gpurun_out> <variant> [<variant> ...] — A/B of kernel variants on the headline bench (GPU box):
# golden check of every non-zero variant, phase profile (variant | 4), then alternating bench runs.
OUT=gpurun_out/$1; shift
mkdir -p $OUT
for v in "$@"; do
  if [ "$v" != "0" ]; then (timeout 200 python tools/variant_check.py $v 2>&1 | tail -1) | tee -a $OUT/check.txt; fi
  timeout 120 python tools/phase_stats.py $((v | 4)) 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('variant $v stats Msamples/s %.1f' % d['Msamples/s']); [print(' ', k, {a:round(b,2) for a,b in d[k].items()}) for k in ('march','block','shade','swaps','loop')]; print('  parts', d.get('parts_share_of_total'))" | tee -a $OUT/stats.txt
done
for rep in 1 2; do for v in "$@"; do
  timeout 120 python bench.py --no-cpu --no-extras --no-roofline --steps 8 --kernel $v 2>/dev/null | python -c "
import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('variant $v', 'Msamples/s', d['value'], 'launch_ms', d['roofline']['launch_ms'])" | tee -a $OUT/ab.txt
done; done
