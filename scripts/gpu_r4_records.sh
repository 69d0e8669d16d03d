#!/bin/bash
# round-4 records in one GPU call: headline rocprofv3 passes, the literal MAC's passes, the driver-style bench line
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
bash scripts/profile_r4.sh headline > gpurun_out/prof_r4_headline.log 2>&1; tail -2 gpurun_out/prof_r4_headline.log
bash scripts/profile_r4.sh literal --mode stream --blocks 2048 > gpurun_out/prof_r4_literal.log 2>&1; tail -2 gpurun_out/prof_r4_literal.log
python bench.py > gpurun_out/r4_bench_cfg3.json 2> gpurun_out/r4_bench_cfg3.err; tail -c 300 gpurun_out/r4_bench_cfg3.json
