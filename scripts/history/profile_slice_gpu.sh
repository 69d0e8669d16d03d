#!/bin/bash
# rocprofv3 kernel stats of one rank of an N-way block-sliced run (compute only); runs on the GPU box
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
N=${1:-8}
OUT=$REPO/gpurun_out/prof_slice$N
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $REPO/bench.py --force-sharded --emulate-world $N --steps 5 --warmup 2 --no-check > $OUT/stats.log 2>&1
tail -1 $OUT/stats.log | cut -c1-200
