#!/bin/bash
# headline bench (no CPU / latency / host legs) for each variant: "default", a library path, or VAR=value[,VAR=value]
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
for round in 1 2; do
  for v in "$@"; do
    ( if [ "$v" != "default" ]; then
        if [[ "$v" == *=* ]]; then IFS=',' read -ra kv <<< "$v"; for x in "${kv[@]}"; do export "$x"; done; else export MCCONV_LIB=$PWD/$v; fi
      fi
      python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-latency --no-host-io 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('[$v]', 'rtf', d['value'], 'ms/step', d['ms_per_step'], 'blocks', d['config']['blocks_per_step'], r['kernel'], r['kernel_avg_ms'], 'parity', d['parity']['rms_err'] if d.get('parity') else None)" )
  done
done
