#!/bin/bash
# rocprofv3 kernel stats of the one-period-per-call (JACK) path; runs on the GPU box
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r1}
OUT=$REPO/gpurun_out/prof_${TAG}_jack
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && export WARM=2000
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $REPO/scripts/latency_probe.py 1000 > $OUT/stats.log 2>&1
tail -1 $OUT/stats.log | cut -c1-200
