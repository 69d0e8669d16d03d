#!/bin/bash
# round 4: the JACK tail with partition 0 in the time domain (default) against the round-3 form (build_ab/lib_fft0.so, -DMCCONV_LAB -DMC_TAIL_FFT0):
# the JACK tests, the in-kernel stamps (build_ab/lib_trace_*.so), then the per-call times back to back and spaced, alternating.
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "jack or parked or period or q8 or speculative or on_process or onProcess or latency" > gpurun_out/tail_td_tests.txt 2>&1
echo "tests rc=$?" >> gpurun_out/tail_td_tests.txt
tail -5 gpurun_out/tail_td_tests.txt
LIBS="${TRACE_LIBS:-trace_td}" bash scripts/gpu_tail_trace.sh | grep -o '\[trace.*' | sed 's/us per.*output issued/ output issued/' | cut -c1-220
bash scripts/gpu_jack_ab.sh default MCCONV_LIB=build_ab/lib_fft0.so > gpurun_out/tail_td_ab.txt 2>&1
cat gpurun_out/tail_td_ab.txt
