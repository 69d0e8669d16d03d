# round 4: randomised runs over the rebuilt 256-frame JACK tail (tests/fuzz/fuzz_q8.py), against oracle.RefCompat: the default library
# (the first look decides the form), then the lab build with each form forced for every period (MCCONV_TAIL_FORM=td / fd)
cd ${GRAFT_REPO_ROOT:-$(pwd)}
{
run() { echo "== [$1] fuzz_q8.py $2"; ( [ "$1" = default ] || export MCCONV_LIB=build_ab/lib_lab.so MCCONV_TAIL_FORM=$1; timeout -k 10 900 python tests/fuzz/fuzz_q8.py $2 2>&1 | grep -v amdgpu.ids | grep 'FAIL\|runs,\|fault\|Error' | cut -c1-260 ); }
run default "72000 60 jack"; run default "1 40"; run default "5200 60 general"; run default "73000 10 jack512"
run td "72000 60 jack"; run td "1 40"; run td "5200 40 general"
run fd "72000 20 jack"; run fd "1 20"
} > gpurun_out/r4_fuzz_tail.txt 2>&1
cat gpurun_out/r4_fuzz_tail.txt
timeout -k 10 300 python -m pytest tests -m gpu -x -q -k "parked_periods_survive" 2>&1 | tail -3
