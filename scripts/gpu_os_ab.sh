#!/bin/bash
# A/B of the overlap-save form against the second-level-transform path on the headline step, plus per-kernel times
set -e
mkdir -p gpurun_out
F="--no-cpu-baseline --no-latency --no-host-io --steps 20 --warmup 5"
MCCONV_OS=0 python bench.py $F > gpurun_out/os_ab_off.json 2> gpurun_out/os_ab_off.err
python bench.py $F > gpurun_out/os_ab_on.json 2> gpurun_out/os_ab_on.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/os_prof -o os -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-latency --no-host-io --no-parity --steps 20 --warmup 5 > $GRAFT_REPO_ROOT/gpurun_out/os_prof.log 2>&1
