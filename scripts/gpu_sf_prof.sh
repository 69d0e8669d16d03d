#!/bin/bash
# per-kernel durations of the single-transform form (bench.py --form single)
cd ${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$PWD/gpurun_out/sf_prof; rm -rf $OUT; mkdir -p $OUT
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $OLDPWD/bench.py --form single --taps 88200 --fft-size 131072 --steps 3 --warmup 1 --prewarm-ms 0 --no-parity --no-latency --no-cpu-baseline "$@" > $OUT/run.log 2>&1)
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/*/*_kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f)))[:8]:
    print('%-40s calls %6s avg %8.2f us  %5.1f %%' % (r['Name'][:40], r['Calls'], float(r['AverageNs']) / 1e3, float(r['Percentage'])))
PY
find $OUT -name "*_kernel_trace.csv" -delete
