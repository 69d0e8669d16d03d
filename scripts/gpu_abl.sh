#!/bin/bash
# Timing-only ablations of k_g2_mac (build_ab/libmcconv_abl{1,2,3}.so: -DG2_ABL=1 no global memory, 2 no butterflies,
# 3 no butterflies and no LDS passes; wrong results by construction, parity skipped) against the shipped build,
# alternating, on one box.   gpu_abl.sh [variants...]
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
VARS=${@:-"default build_ab/libmcconv_abl1.so build_ab/libmcconv_abl2.so build_ab/libmcconv_abl3.so"}
for round in 1 2 3; do
  for v in $VARS; do
    ( if [ "$v" != "default" ]; then
        if [[ "$v" == *=* ]]; then IFS=',' read -ra kv <<< "$v"; for x in "${kv[@]}"; do export "$x"; done; else export MCCONV_LIB=$PWD/$v; fi
      fi
      timeout -k 10 120 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-latency --no-host-io --no-parity 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('[$v]', 'rtf', d['value'], 'ms/step', d['ms_per_step'], r['kernel'], r['kernel_avg_ms'], 'frac', r['frac'])" ) || exit 1
  done
done
