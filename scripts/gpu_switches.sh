#!/bin/bash
# the GPU suite under each measurement switch (all of them select product paths)
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
for v in ${SWITCHES:-"MCCONV_HOST_OUT_DIRECT=0" "MCCONV_BAR_IO=0" "MCCONV_NO_PARK=1" "MCCONV_G2_WIDE=1" "MCCONV_CORR_RIDE=0" "MCCONV_FFT2_FUSED=0" "MCCONV_FFT2=0" "MCCONV_INV_WET=0" "MCCONV_NO_SPECULATE=1" "MCCONV_NO_SPIN=1" "MCCONV_FFA_LEVELS=0" "MCCONV_FUSE_OUT=0" "MCCONV_TAGGED_IO=0" "MCCONV_TAGGED_IO=2" "MCCONV_TD_FFT=0" "MCCONV_FUSE_DROP=0" "MCCONV_DROP_AHEAD=0" "MCCONV_CARRY_DROP=0" "MCCONV_HTAIL=0" "MCCONV_G2_DUO=1"}; do
  ( export "$v"; timeout -k 10 600 python -m pytest tests -m gpu -q --tb=line -x > gpurun_out/t_sw.log 2>&1; echo "[$v] rc=$? $(tail -1 gpurun_out/t_sw.log)" | tee -a gpurun_out/switches.txt; grep -E "^FAILED|Error" gpurun_out/t_sw.log | head -3 )
done
