#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
for v in "$@"; do
  ( if [ "$v" != "default" ]; then export "$v"; fi
    OUT=$PWD/gpurun_out/emu_$(echo $v | tr '/=' '__'); rm -rf $OUT; mkdir -p $OUT
    (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $OLDPWD/bench.py --steps 10 --warmup 3 --prewarm-ms 0 --force-sharded --emulate-world 8 --layouts blocks --no-check > $OUT/run.log 2>&1)
    python3 - "$OUT" "$v" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/*/*_kernel_stats.csv')[0]
print('[%s]' % sys.argv[2])
for r in list(csv.DictReader(open(f)))[:7]:
    print('   %-40s calls %5s avg %8.2f us' % (r['Name'][:40], r['Calls'], float(r['AverageNs']) / 1e3))
PY
    find $OUT -name "*_kernel_trace.csv" -delete )
done
