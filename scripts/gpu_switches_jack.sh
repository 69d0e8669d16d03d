#!/bin/bash
# the JACK-path tests of the GPU suite under every switch that touches that path (lab build; "default" = the default library) - after the
# 256-frame tail was rebuilt at the end of round 4
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
: > gpurun_out/switches_jack.txt
for v in "default" "lab" "MCCONV_BAR_IO=0" "MCCONV_NO_PARK=1" "MCCONV_NO_SPIN=1" "MCCONV_PARK_MS=5" "MCCONV_TAGGED_IO=0" "MCCONV_TAGGED_IO=2" "MCCONV_NO_SPECULATE=1" "MCCONV_TAIL_FORM=td" "MCCONV_TAIL_FORM=fd" "MCCONV_TAIL_FORM=td,MCCONV_BAR_IO=0" "MCCONV_TAIL_FORM=td,MCCONV_NO_PARK=1" "MCCONV_CARRY_DROP=0" "MCCONV_TD_FFT=0" "MCCONV_FFA_LEVELS=0"; do
  ( if [ "$v" != "default" ]; then export MCCONV_LIB=$PWD/build_ab/lib_lab.so; fi
    if [ "$v" != "default" ] && [ "$v" != "lab" ]; then IFS=',' read -ra kv <<< "$v"; for x in "${kv[@]}"; do export "$x"; done; fi
    timeout -k 10 600 python -m pytest tests -m gpu -q --tb=line -x -p no:cacheprovider -k "jack or parked or period or q8 or speculative or on_process or onProcess or latency or smoke or golden or fuzz" > gpurun_out/t_sw.log 2>&1; echo "[$v] rc=$? $(tail -1 gpurun_out/t_sw.log)" | tee -a gpurun_out/switches_jack.txt; grep -E "^FAILED|Error" gpurun_out/t_sw.log | head -3 )
done
