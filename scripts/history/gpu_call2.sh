#!/bin/bash
set -o pipefail
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
bash scripts/profile_r2.sh headline || exit 1
echo "== config 2 / 5 bench lines"
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --taps 88200 --fft-size 131072 --no-latency --no-host-io --cpu-seconds 5 > gpurun_out/bench_cfg2.json 2> gpurun_out/bench_cfg2.err; echo "cfg2 rc=$?"
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --taps 1323000 --fft-size 2097152 --no-latency --no-host-io --cpu-seconds 5 > gpurun_out/bench_cfg5_fp32.json 2> gpurun_out/bench_cfg5_fp32.err; echo "cfg5 fp32 rc=$?"
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --taps 1323000 --fft-size 2097152 --precision fp16 --blocks 2048 --no-latency --no-host-io --no-cpu-baseline > gpurun_out/bench_cfg5_fp16.json 2> gpurun_out/bench_cfg5_fp16.err; echo "cfg5 fp16 rc=$?"
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --taps 1323000 --fft-size 2097152 --mode stream --blocks 2048 --no-latency --no-host-io --no-cpu-baseline > gpurun_out/bench_cfg5_stream32.json 2> gpurun_out/bench_cfg5_stream32.err; echo "cfg5 stream rc=$?"
echo "== multi-GPU rehearsal on one GPU (rank 0 of 8)"
timeout -k 10 400 python bench.py --steps 10 --warmup 3 --force-sharded --emulate-world 8 > gpurun_out/bench_emu8.json 2> gpurun_out/bench_emu8.err; echo "emu8 rc=$?"
timeout -k 10 400 python bench.py --steps 10 --warmup 3 --channels 8 --no-latency --no-host-io --no-cpu-baseline > gpurun_out/bench_ch8.json 2> gpurun_out/bench_ch8.err; echo "ch8 rc=$?"
for f in cfg2 cfg5_fp32 cfg5_fp16 cfg5_stream32 emu8 ch8; do echo "--- $f"; cut -c1-700 gpurun_out/bench_$f.json; tail -3 gpurun_out/bench_$f.err; done
echo "== full GPU suite"
python -m pytest tests -m gpu -q --tb=short > gpurun_out/t_full.log 2>&1; echo "pytest rc=$?"
tail -15 gpurun_out/t_full.log
