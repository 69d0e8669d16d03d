# the 256-frame JACK call with a whole real period (5805 us) of idle time between calls, as under jackd at 44.1 kHz: default library against round 3's tail
cd ${GRAFT_REPO_ROOT:-$(pwd)}
for round in 1 2; do
  for v in default build_ab/lib_fft0.so; do
    if [ "$v" = default ]; then r=$(python scripts/jack_loop.py 500 5805 256 2>/dev/null | tail -1); else r=$(MCCONV_LIB=$v python scripts/jack_loop.py 500 5805 256 2>/dev/null | tail -1); fi
    echo "[$v] $r"
  done
done | tee gpurun_out/jack_real_spacing.txt
