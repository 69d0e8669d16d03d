#!/bin/bash
# sweep MCCONV_NCHUNK for the single-block path; prints rocprof avg kernel times
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for NC in "$@"; do
  export MCCONV_NCHUNK=$NC
  rm -rf $REPO/gpurun_out/prof_sweep
  rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_sweep -- python3 $REPO/scripts/latency_probe.py 300 > $REPO/gpurun_out/prof_sweep.log 2>&1
  python3 - <<PY
import csv,glob
f=glob.glob("$REPO/gpurun_out/prof_sweep/*/*_kernel_stats.csv")[0]
out=[]
for r in csv.DictReader(open(f)):
    n=r["Name"].split("(")[0]
    if n.startswith("k_mac") or n.startswith("void k_mac") or "k_tail" in n:
        out.append("%s %.2fus"%(n[-22:], float(r["AverageNs"])/1e3))
print("nchunk=$NC", " | ".join(out), open("$REPO/gpurun_out/prof_sweep.log").read().strip().splitlines()[-1][:80])
PY
done
