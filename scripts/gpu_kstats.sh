#!/bin/bash
# per-kernel average durations of the headline step for each variant (library path or VAR=value or "default")
cd ${GRAFT_REPO_ROOT:-$(pwd)}
for v in "$@"; do
  ( if [ "$v" != "default" ]; then if [[ "$v" == *=* ]]; then export "$v"; else export MCCONV_LIB=$PWD/$v; fi; fi
    OUT=$PWD/gpurun_out/ks_$(echo $v | tr '/=' '__'); rm -rf $OUT; mkdir -p $OUT
    (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $OLDPWD/bench.py --steps 20 --warmup 5 --prewarm-ms 100 --no-cpu-baseline --no-latency --no-host-io --no-parity --no-literal-mac $BENCH_ARGS > $OUT/run.log 2>&1)
    python3 - "$OUT" "$v" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/*/*_kernel_stats.csv')[0]
print('[%s]' % sys.argv[2], ' '.join('%s %.1f' % (r['Name'].split('(')[0].replace('void ','')[:14], float(r["AverageNs"]) / 1e3) for r in list(csv.DictReader(open(f)))[:9]))
PY
    find $OUT -name "*_kernel_trace.csv" -delete )
done
