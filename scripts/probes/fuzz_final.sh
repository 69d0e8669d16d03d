# round 4, final library: a last pass of the randomised runs (tests/fuzz/) over every mode, against oracle.RefCompat
cd ${GRAFT_REPO_ROOT:-$(pwd)}
{
run() { echo "== $*"; timeout -k 10 1000 python "$@" 2>&1 | grep -v amdgpu.ids | grep 'FAIL\|runs\b.*above\|runs {\|fault\|Error' | cut -c1-260; }
run tests/fuzz/fuzz_q8.py 91000 150 general
run tests/fuzz/fuzz_q8.py 92000 100 jack
run tests/fuzz/fuzz_q8.py 93000 100
run tests/fuzz/fuzz_q8.py 94000 20 jack512
run tests/fuzz/fuzz_q8.py 95000 20 jack1024
run tests/fuzz/fuzz_q8.py 96000 25 long
run tests/fuzz/fuzz_shards.py 97000 40
} > gpurun_out/r4_fuzz_final.txt 2>&1
cat gpurun_out/r4_fuzz_final.txt
