#!/bin/bash
# first GPU call of the round: the new tests, the bench line, the counter list
set -o pipefail
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -k "headline or any_grid or pinned_host or config4_partition or q4_output" > gpurun_out/t_new.log 2>&1
echo "pytest new rc=$?" | tee -a gpurun_out/t_new.log
tail -25 gpurun_out/t_new.log
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_r2a.json 2> gpurun_out/bench_r2a.err
echo "bench rc=$?"
cut -c1-3000 gpurun_out/bench_r2a.json
tail -5 gpurun_out/bench_r2a.err
(cd /tmp && export TMPDIR=/tmp && rocprofv3 -L > $OLDPWD/gpurun_out/counters.txt 2>&1)
grep -c . gpurun_out/counters.txt
