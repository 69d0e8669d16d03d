#!/bin/bash
# rocprofv3 passes of the JACK path (scripts/jack_loop.py): kernel stats, then FETCH_SIZE / WRITE_SIZE / L2 hit counters
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_r2_jack
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
echo "scripts/jack_loop.py 3000" > $OUT/command.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $REPO/scripts/jack_loop.py 3000 > $OUT/stats.log 2>&1 || { tail -5 $OUT/stats.log; exit 1; }
for p in "pmc_fetch:FETCH_SIZE" "pmc_write:WRITE_SIZE" "pmc_tcc:TCC_HIT_sum TCC_MISS_sum"; do
  n=${p%%:*}; c=${p#*:}
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/$n -- python3 $REPO/scripts/jack_loop.py 600 > $OUT/$n.log 2>&1 || echo "pass $n failed"
  find $OUT/$n -name "*_kernel_trace.csv" -delete
done
find $OUT -name "*_kernel_trace.csv" -delete
tail -1 $OUT/stats.log
