#!/bin/bash
# k_g2_mac with and without the asymmetric issue priority (build_ab/libmcconv_noprio.so: -DG2_PRIO=0), alternating, over batch lengths and configs
cd ${GRAFT_REPO_ROOT:-$(pwd)}
run() {  # label, lib-or-default, bench args
  ( if [ "$2" != "default" ]; then export MCCONV_LIB=$PWD/$2; fi
    python bench.py --steps ${STEPS:-60} --warmup 10 --no-cpu-baseline --no-latency --no-host-io --no-parity ${@:3} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('[$1 | $2]', 'rtf %.0f ms/step %.4f kernel %s %.1f us' % (d['value'], d['ms_per_step'], r.get('kernel'), 1e3 * r.get('kernel_avg_ms', 0)))" )
}
for rep in 1 2; do
  for lib in default build_ab/libmcconv_noprio.so; do
    run "cfg3 T=129296" $lib
    run "cfg3 T=32320" $lib --blocks 32320
    run "cfg3 T=8192" $lib --blocks 8192
    run "cfg5 (30 s IRs)" $lib --taps 1323000 --fft-size 2097152
    run "cfg2 (2 s IR)" $lib --taps 88200 --fft-size 131072
    run "shipped" $lib --shipped-defaults
  done
done
