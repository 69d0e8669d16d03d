# what the frequency-domain form of the 256-frame tail takes from the period's arrival to its output: cold (as it runs) against warm (a dry pass
# of the stretch first; measurement build -DMC_FD_WARM), in-kernel stamps, the form forced on spaced parked periods and on calls back to back
cd ${GRAFT_REPO_ROOT:-$(pwd)}
for lib in trace_td trace_fdwarm trace_td trace_fdwarm; do
  for gap in 0 500; do
    echo "[$lib fd gap $gap] $(MCCONV_TAIL_FORM=fd MCCONV_LIB=build_ab/lib_$lib.so python scripts/jack_loop.py 2000 $gap 256 2>&1 | tail -2 | tr '\n' ' ' | grep -o 'output issued[^,]*')"
  done
done | tee gpurun_out/fd_warm.txt
