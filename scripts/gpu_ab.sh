#!/bin/bash
# A/B on the headline bench: each argument is "default", a library path (MCCONV_LIB) or VAR=value settings
# (comma-separated); three rounds, alternating.   gpu_ab.sh [--blocks N] default MCCONV_G2_WIDE=1 ...
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
EXTRA=""
if [ "$1" = "--blocks" ]; then EXTRA="--blocks $2"; shift 2; fi
for round in 1 2 3; do
  for v in "$@"; do
    ( if [ "$v" != "default" ]; then
        if [[ "$v" == *=* ]]; then IFS=',' read -ra kv <<< "$v"; for x in "${kv[@]}"; do export "$x"; done; else export MCCONV_LIB=$PWD/$v; fi
      fi
      python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-latency --no-host-io $EXTRA 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('[$v]', 'rtf', d['value'], 'ms/step', d['ms_per_step'], r['kernel'], r['kernel_avg_ms'], 'frac', r['frac'], 'parity', d['parity']['rms_err'] if d.get('parity') else None)" )
  done
done
