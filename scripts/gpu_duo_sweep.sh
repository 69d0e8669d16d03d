#!/bin/bash
# k_g2_duo against k_g2_mac over batch lengths (chunks per bin) and for the 30 s IRs of config 5, same box, alternating.
cd ${GRAFT_REPO_ROOT:-$(pwd)}
run() {  # label, env, bench args
  ( export $2; timeout -k 10 120 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-latency --no-host-io --no-parity $3 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', '$2', 'rtf', d['value'], 'ms/step', d['ms_per_step'], r['kernel'], r['kernel_avg_ms'])" )
}
for T in 12928 19392 32320 64640 129296; do
  for rep in 1 2; do
    run "T=$T" MCCONV_G2_DUO=1 "--blocks $T"
    run "T=$T" MCCONV_G2_DUO=0 "--blocks $T"
  done
done
for rep in 1 2; do
  run "cfg5" MCCONV_G2_DUO=1 "--taps 1323000 --fft-size 2097152"
  run "cfg5" MCCONV_G2_DUO=0 "--taps 1323000 --fft-size 2097152"
  run "cfg2" MCCONV_G2_DUO=1 "--taps 88200 --fft-size 131072"
  run "cfg2" MCCONV_G2_DUO=0 "--taps 88200 --fft-size 131072"
done
