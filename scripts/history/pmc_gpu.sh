#!/bin/bash
# usage: pmc_gpu.sh <tag> "<counters>" [bench args]; runs on the GPU box
set -o pipefail
TAG=$1; CTRS=$2; shift 2
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d $OUT -- python3 $REPO/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-latency "$@" > $OUT/run.log 2>&1 || { tail -5 $OUT/run.log; exit 1; }
python3 - <<PY
import csv,glob,collections
f=glob.glob('$OUT/*/*_counter_collection.csv')[0]
agg=collections.defaultdict(list)
dur=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    k=r['Kernel_Name'].split('(')[0]
    agg[(k,r['Counter_Name'])].append(float(r['Counter_Value']))
    dur[k].append(int(r['End_Timestamp'])-int(r['Start_Timestamp']))
for (k,c),v in sorted(agg.items()):
    if k.startswith('k_mac'):
        print(k,c,'mean=%.4g'%(sum(v)/len(v)),'dur_us=%.1f'%(sum(dur[k])/len(dur[k])/1e3))
PY
grep '"metric"' $OUT/run.log | cut -c1-160
