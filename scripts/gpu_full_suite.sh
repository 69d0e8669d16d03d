#!/bin/bash
# the GPU suite, smoke() and a short JACK A/B in one call
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/full_suite.txt 2>&1
rc=$?
tail -4 gpurun_out/full_suite.txt
[ $rc -eq 0 ] || exit $rc
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash scripts/gpu_jack_ab.sh default > gpurun_out/jack_default.txt 2>&1; cat gpurun_out/jack_default.txt
