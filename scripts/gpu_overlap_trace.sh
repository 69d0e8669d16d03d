#!/bin/bash
# Do the launches of two engines on two streams overlap on the GPU, and what does it buy?  (kernel trace of `bench.py --channels 4`)
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/overlap
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
A="--steps 10 --warmup 2 --prewarm-ms 100 --no-cpu-baseline --no-latency --no-host-io --no-parity"
timeout -k 10 120 python3 $REPO/bench.py $A --channels 2 > $OUT/ch2.json 2> $OUT/ch2.err
timeout -k 10 120 python3 $REPO/bench.py $A --channels 4 > $OUT/ch4.json 2> $OUT/ch4.err
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $REPO/bench.py $A --channels 4 > $OUT/trace.log 2>&1
python3 - <<PY
import csv, glob, json
for f in ("ch2","ch4"):
    d=json.loads(open("$OUT/%s.json"%f).read().strip().splitlines()[-1]); print(f, d["value"], d["ms_per_step"])
fn=glob.glob("$OUT/trace/**/*_kernel_trace.csv", recursive=True)[0]
rows=list(csv.DictReader(open(fn)))
rows=[r for r in rows if r["Kernel_Name"].startswith(("k_fwd","k_g2_mac","k_inv_wet"))]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
t0=int(rows[-40]["Start_Timestamp"])
with open("$OUT/timeline.txt","w") as o:
    for r in rows[-40:]:
        o.write("%-12s q%s  %8.1f -> %8.1f us  (%.1f)\n"%(r["Kernel_Name"][:12], r.get("Queue_Id","?"), (int(r["Start_Timestamp"])-t0)/1e3,(int(r["End_Timestamp"])-t0)/1e3,(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3))
print(open("$OUT/timeline.txt").read())
PY
find $OUT/trace -name "*.csv" -size +1M -delete
