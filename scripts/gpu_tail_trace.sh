#!/bin/bash
# in-kernel stamps of the JACK tail (MC_JACK_TRACE builds): time-domain partition 0 against the round-3 form, back to back and spaced
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
for lib in ${LIBS:-trace_td trace_fft0 trace_td trace_fft0}; do
  for gap in 0 500; do
    echo "[$lib gap $gap] $(MCCONV_LIB=build_ab/lib_$lib.so python scripts/jack_loop.py 2000 $gap 256 2>&1 | tail -2 | tr '\n' ' ')"
  done
done > gpurun_out/tail_trace.txt 2>&1
cat gpurun_out/tail_trace.txt
