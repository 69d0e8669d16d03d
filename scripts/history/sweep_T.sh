#!/bin/bash
for T in 16 32 64 128 256 512; do
  for M in resident stream; do
    python bench.py --no-cpu-baseline --no-latency --steps 40 --warmup 60 --blocks $T --mode $M 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('T=$T', '$M', 'rtf', d['value'], 'ms/step', d['ms_per_step'], 'kernel', d['roofline']['kernel'], d['roofline']['kernel_avg_ms'])"
  done
done
