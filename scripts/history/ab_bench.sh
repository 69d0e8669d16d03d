#!/bin/bash
# alternate two environment settings of bench.py a few times (run-to-run noise is ~1 %)
for i in 1 2 3; do
  for v in "" "$1"; do
    env $v python bench.py --no-cpu-baseline --no-latency --steps 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$v]', d['value'], d['ms_per_step'], d['roofline']['kernel_avg_ms'])"
  done
done
