#!/bin/bash
# k_drop_fft at different register caps (libmcconv_eu<N>.so built with -DDF_EU=N): kernel table of the shipped operating point
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/q8v
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
A="--steps 20 --warmup 3 --prewarm-ms 100 --no-cpu-baseline --no-latency --no-host-io --no-parity --shipped-defaults"
for v in "$@"; do
  export MCCONV_LIB=$REPO/cuda_audio_amd/libmcconv_$v.so
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$v -- python3 $REPO/bench.py $A > $OUT/$v.json 2> $OUT/$v.err || { tail -5 $OUT/$v.err; exit 1; }
  python3 - <<PY
import csv, glob, json
d=json.loads(open("$OUT/$v.json").read().strip().splitlines()[-1]); print("$v rtf", d["value"], "ms/step", d["ms_per_step"])
fn=glob.glob("$OUT/$v/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(fn)))[:6]:
    if "drop" in r["Name"] or "post" in r["Name"]: print("   %-40s avg %10.1f us"%(r["Name"][:40], float(r["AverageNs"])/1e3))
PY
  find $OUT/$v -name "*_kernel_trace.csv" -delete
done
