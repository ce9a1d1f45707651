for lv in "6,3" "7,2" "5,4" "8,1" "6,2,1"; do
  for rep in 1 2; do
    CHUNKY_WIDE_LEVELS=$lv timeout 120 python bench.py --no-cpu --no-extras --steps 6 --kernel 128 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('levels $lv', d['value'], 'launch_ms', d['roofline']['launch_ms'], d['roofline']['kernel'], flush=True)"
  done
done
