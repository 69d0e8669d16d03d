#!/bin/bash
# Runs on the GPU box (via gpurun): per-kernel average durations of the headline bench step (rocprofv3 --stats only).
set -o pipefail
TAG=${1:-k}
EXTRA="${@:2}"  # further bench.py arguments
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/kstats_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $REPO/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-latency --no-check $EXTRA > $OUT/run.log 2>&1 || exit 1
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/*/*_kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f)))[:9]:
    print(r['Name'][:40].ljust(40), r['Calls'].rjust(5), '%9.1f us' % (float(r['AverageNs']) / 1e3), r['Percentage'])
PY
