#!/bin/bash
# per-launch durations of the three overlap-save passes in dispatch order (is the spread between launches or between runs?)
cd ${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$PWD/gpurun_out/os_trace; rm -rf $OUT; mkdir -p $OUT
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $OLDPWD/bench.py --steps 24 --warmup 4 --prewarm-ms 100 --no-cpu-baseline --no-latency --no-host-io --no-parity --no-literal-mac > $OUT/run.log 2>&1)
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/*/*_kernel_trace.csv')[0]
rows = [r for r in csv.DictReader(open(f))]
for name in ('k_os_cols', 'k_os_rows', 'k_os_out'):
    d = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in rows if r['Kernel_Name'].startswith(name + '(') or r['Kernel_Name'].startswith('void ' + name + '(')]
    d = d[-28:]
    print(name, ' '.join('%.0f' % x for x in d))
PY
rm -rf $OUT
