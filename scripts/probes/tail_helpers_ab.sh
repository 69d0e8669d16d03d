cd ${GRAFT_REPO_ROOT:-$(pwd)}
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "jack or parked or period or q8 or speculative or on_process or onProcess or latency" > gpurun_out/tail_td_tests.txt 2>&1
echo "tests rc=$?"; tail -3 gpurun_out/tail_td_tests.txt
CALLS=2000 bash scripts/gpu_jack_p50.sh default build_ab/lib_td_a.so build_ab/lib_fft0.so
LIBS=trace_td bash scripts/gpu_tail_trace.sh | grep -o "\[trace.*" | sed "s/us per.*output issued/ output issued/" | cut -c1-200
bash scripts/gpu_jack_ab.sh default build_ab/lib_td_a.so 2>&1 | cut -c1-120
