#!/bin/bash
# the GPU suite under each measurement switch, on the LAB build (build_ab/lib_lab.so: scripts/build_variant.sh lab -DMCCONV_LAB) -
# the default library reads ten switches only.  "default" = the default library with no switch.
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
: > gpurun_out/switches.txt
for v in ${SWITCHES:-"default" "lab" "MCCONV_OS=0" "MCCONV_OS_SIDE=0" "MCCONV_HOST_OUT_DIRECT=0" "MCCONV_BAR_IO=0" "MCCONV_NO_PARK=1" "MCCONV_G2_WIDE=1,MCCONV_OS=0" "MCCONV_CORR_RIDE=0" "MCCONV_FFT2_FUSED=0,MCCONV_OS=0" "MCCONV_FFT2=0" "MCCONV_INV_WET=0" "MCCONV_NO_SPECULATE=1" "MCCONV_NO_SPIN=1" "MCCONV_FFA_LEVELS=0" "MCCONV_FUSE_OUT=0" "MCCONV_TAGGED_IO=0" "MCCONV_TAGGED_IO=2" "MCCONV_TD_FFT=0" "MCCONV_FUSE_DROP=0" "MCCONV_DROP_AHEAD=0" "MCCONV_CARRY_DROP=0" "MCCONV_HTAIL=0" "MCCONV_G2_DUO=1,MCCONV_OS=0"}; do
  ( if [ "$v" != "default" ]; then export MCCONV_LIB=$PWD/build_ab/lib_lab.so; fi
    if [ "$v" != "default" ] && [ "$v" != "lab" ]; then IFS=',' read -ra kv <<< "$v"; for x in "${kv[@]}"; do export "$x"; done; fi
    timeout -k 10 900 python -m pytest tests -m gpu -q --tb=line -x -p no:cacheprovider > gpurun_out/t_sw.log 2>&1; echo "[$v] rc=$? $(tail -1 gpurun_out/t_sw.log)" | tee -a gpurun_out/switches.txt; grep -E "^FAILED|Error" gpurun_out/t_sw.log | head -3 )
done
