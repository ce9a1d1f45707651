export CHUNKY_ORACLE_NO_BUILD=1
(timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_timed_kernels.py -m gpu -q -x --timeout 600 2>&1 | tail -3)
bash tools/ab.sh "base default" bench benchmark indoor entities
cd /tmp && export TMPDIR=/tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu --no-roofline --no-extras --steps 6 > /tmp/kt.log 2>&1; find /tmp/kt -name "*kernel_stats.csv" | head -1 | xargs head -4
