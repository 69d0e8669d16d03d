cd ${GRAFT_REPO_ROOT:-$(pwd)}
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "jack or parked or period or q8 or speculative or on_process or onProcess or latency" > gpurun_out/tail_td_tests.txt 2>&1
echo "tests rc=$?"; tail -3 gpurun_out/tail_td_tests.txt
PERIODS="512 1024" bash scripts/gpu_jack_ab.sh default MCCONV_LIB=build_ab/lib_td_b.so 2>&1 | cut -c1-260
