#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
for T in "$@"; do
  python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-latency --no-host-io --no-parity --no-literal-mac --blocks $T 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('T=$T', 'rtf', d['value'], 'ms/step', d['ms_per_step'], 'ns/block', round(d['ms_per_step']*1e6/$T,2), r['kernel'], r['kernel_avg_ms'])"
done
