#!/bin/bash
# Q8 regime: the tests that cover it, then the shipped operating point under rocprofv3 --stats (kernel table to gpurun_out/q8/)
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/q8
mkdir -p $OUT
cd $REPO
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "q8 or tail_drop or predelay or shipped or Q8" > $OUT/tests.log 2>&1 || { tail -30 $OUT/tests.log; exit 1; }
tail -2 $OUT/tests.log
cd /tmp && export TMPDIR=/tmp
A="--steps 20 --warmup 3 --prewarm-ms 100 --no-cpu-baseline --no-latency --no-host-io --shipped-defaults $@"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $REPO/bench.py $A > $OUT/bench.json 2> $OUT/bench.err || { tail -5 $OUT/bench.err; exit 1; }
python3 - <<PY
import csv, glob, json
d=json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1]); print("rtf", d["value"], "ms/step", d["ms_per_step"], "parity", d.get("parity",{}).get("rms_err"), d.get("parity",{}).get("ok"))
fn=glob.glob("$OUT/stats/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(fn)))[:8]:
    print("%-60s calls %6s avg %10.1f us  %5.1f%%"%(r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3, float(r["Percentage"])))
PY
find $OUT/stats -name "*_kernel_trace.csv" -delete
