#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel stats + HBM traffic counters
# for the bench command.  Outputs under gpurun_out/prof_<tag>/.
set -o pipefail
TAG=${1:-r1}
MODE=${2:-resident}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_${TAG}_${MODE}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
EXTRA=""; [ "$MODE" = "stream" ] && EXTRA="--blocks 2048"
ARGS="$REPO/bench.py --steps 10 --warmup 2 --mode $MODE --no-cpu-baseline --no-latency $EXTRA"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ARGS > $OUT/stats.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ARGS > $OUT/pmc_fetch.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ARGS > $OUT/pmc_write.log 2>&1 || exit 1
find $OUT -name "*.csv" | head -20
