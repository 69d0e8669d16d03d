# 512- and 1024-frame JACK periods: the default library against build_ab/lib_td_c.so (the committed state before a change of k_tailp), alternating; the period tests first
cd ${GRAFT_REPO_ROOT:-$(pwd)}
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "period or jack512 or parked" > gpurun_out/tailp_tests.txt 2>&1
echo "tests rc=$?"; tail -2 gpurun_out/tailp_tests.txt
PERIODS="512 1024" bash scripts/gpu_jack_ab.sh default MCCONV_LIB=build_ab/lib_td_c.so default MCCONV_LIB=build_ab/lib_td_c.so 2>&1 | cut -c1-250
