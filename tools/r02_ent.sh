#!/bin/bash
# GPU box: entity-scene (BASELINE configs[4]) check: parity on the small entity goldens, config bench of the default build and of tuning builds
timeout 200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_timed_kernels.py -m gpu -x -q --timeout 100 -k "entit" 2>&1 | tail -1
timeout 300 python tools/config_bench.py entities 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('default', round(d['Msamples/s'],1), d['rows_bit_identical_to_oracle'])"
for v in "$@"; do
  CHUNKY_HIP_LIB=$PWD/.variants/libchunky_hip_$v.so timeout 300 python tools/config_bench.py entities 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$v', round(d['Msamples/s'],1), d['rows_bit_identical_to_oracle'])"
done
