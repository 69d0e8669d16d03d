#!/bin/bash
# the GPU suite, smoke() and the JACK A/B (default library, round-3 tail build_ab/lib_fft0.so) in one call
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/full_suite.txt 2>&1
rc=$?
tail -4 gpurun_out/full_suite.txt
[ $rc -eq 0 ] || exit $rc
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
CALLS=2000 bash scripts/gpu_jack_p50.sh default build_ab/lib_fft0.so
bash scripts/gpu_jack_ab.sh default MCCONV_LIB=build_ab/lib_fft0.so > gpurun_out/jack_ab.txt 2>&1; cut -c1-250 gpurun_out/jack_ab.txt
