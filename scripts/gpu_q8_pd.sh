#!/bin/bash
# Q8 regime, batches: frequency-domain form (k_drop_fft + k_post<3>) against the time-domain tiles (MCCONV_TD_FFT=0) over predelays
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/q8pd
mkdir -p $OUT
cd $REPO
A="--steps 10 --warmup 2 --prewarm-ms 100 --no-cpu-baseline --no-latency --no-host-io --fft-size 131072 --taps 130048 --same-ir"
for pd in 1024 4160 8192; do
  for v in "MCCONV_TD_FFT=1" "MCCONV_TD_FFT=1 MCCONV_HTAIL=0" "MCCONV_TD_FFT=0"; do
    env $v timeout -k 10 200 python3 bench.py $A --predelay $pd > $OUT/o.json 2> $OUT/o.err || { tail -5 $OUT/o.err; exit 1; }
    python3 - <<PY
import json
d=json.loads(open("$OUT/o.json").read().strip().splitlines()[-1]); print("pd $pd  %-32s rtf %9.0f  ms/step %8.4f  parity %.3e %s"%("$v", d["value"], d["ms_per_step"], d["parity"]["rms_err"], d["parity"]["ok"]))
PY
  done
done
