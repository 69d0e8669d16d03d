#!/bin/bash
# Clock and power while the headline bench runs back to back (rocm-smi polled every ~0.1 s beside a long bench run).
#   gpu_power.sh [VAR=value ...]
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
for kv in "$@"; do export "$kv"; done
( timeout -k 5 60 python bench.py --steps 6000 --warmup 5 --no-cpu-baseline --no-latency --no-host-io --no-parity > gpurun_out/power_bench.json 2>/dev/null ) &
BP=$!
sleep 6
for i in $(seq 1 25); do
  rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "sclk|Socket Power|Average|mclk|fclk|junction" | tr '\n' ' ' | sed 's/  */ /g'
  echo
  sleep 0.15
done
wait $BP
python -c "
import json; d=json.load(open('gpurun_out/power_bench.json')); r=d['roofline']; print('rtf', d['value'], 'ms/step', d['ms_per_step'], r['kernel'], r['kernel_avg_ms'])"
