#!/bin/bash
# A/B of kernel variants: alternate libraries (MCCONV_LIB) on the headline bench, a few rounds each
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
for round in 1 2 3; do
  for lib in "$@"; do
    if [ "$lib" = "default" ]; then unset MCCONV_LIB; else export MCCONV_LIB=$PWD/$lib; fi
    python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-latency --no-host-io 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('[$lib]', 'rtf', d['value'], 'ms/step', d['ms_per_step'], r['kernel'], r['kernel_avg_ms'], 'parity', d['parity']['rms_err'] if d.get('parity') else None)"
  done
done
