#!/bin/bash
set -o pipefail
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
bash scripts/profile_r2.sh headline || exit 1

MCCONV_G2_WIDE=1 MCCONV_CORR_RIDE=0 bash scripts/profile_r2.sh wide || true
bash scripts/profile_jack.sh || true
echo "== bench lines"
run() { n=$1; shift; timeout -k 10 400 python bench.py "$@" > gpurun_out/bench_$n.json 2> gpurun_out/bench_$n.err; echo "$n rc=$?"; }
run cfg3 --steps 20 --warmup 5
run cfg3_T32320 --steps 20 --warmup 5 --blocks 32320 --no-latency --no-host-io --no-cpu-baseline
run cfg3_long --steps 200 --warmup 20
run cfg3_noprewarm --steps 20 --warmup 5 --prewarm-ms 0 --no-latency --no-host-io --no-cpu-baseline
run cfg3_noprewarm_w200 --steps 20 --warmup 200 --prewarm-ms 0 --no-latency --no-host-io --no-cpu-baseline
run cfg2 --steps 20 --warmup 5 --taps 88200 --fft-size 131072 --no-latency --no-host-io --cpu-seconds 5
run cfg2_single --form single --steps 10 --warmup 2 --taps 88200 --fft-size 131072
run cfg5_fp32 --steps 20 --warmup 5 --taps 1323000 --fft-size 2097152 --no-latency --no-host-io --cpu-seconds 5
MCCONV_FFT2_FUSED=0 run cfg5_split --steps 20 --warmup 5 --taps 1323000 --fft-size 2097152 --no-latency --no-host-io --no-cpu-baseline
run ir1s --steps 20 --warmup 5 --taps 44100 --fft-size 65536 --no-latency --no-host-io --no-cpu-baseline
run cfg5_fp16 --steps 20 --warmup 5 --taps 1323000 --fft-size 2097152 --precision fp16 --blocks 2048 --no-latency --no-host-io --no-cpu-baseline
run cfg5_stream32 --steps 20 --warmup 5 --taps 1323000 --fft-size 2097152 --mode stream --blocks 2048 --no-latency --no-host-io --no-cpu-baseline
MCCONV_FFT2=0 MCCONV_FFA_LEVELS=0 run cfg3_direct_mac --steps 10 --warmup 3 --blocks 8192 --no-latency --no-host-io --no-cpu-baseline
run emu8 --steps 10 --warmup 3 --force-sharded --emulate-world 8
run emu2 --steps 10 --warmup 3 --force-sharded --emulate-world 2
run ch8 --steps 10 --warmup 3 --channels 8 --no-latency --no-host-io --no-cpu-baseline
run ch8_emu8 --steps 6 --warmup 2 --channels 8 --force-sharded --emulate-world 8
for f in gpurun_out/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); r=d.get('roofline') or {}
    print(sys.argv[1].split('bench_')[1], d['value'], d['ms_per_step'], r.get('kernel'), r.get('kernel_avg_ms'), r.get('frac'), (d.get('parity') or {}).get('rms_err'), (d.get('north_star_layout') or {}).get('value'), d.get('sharded_check'))
except Exception as e: print(sys.argv[1], 'ERR', e)
PY
done
