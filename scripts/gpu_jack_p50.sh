#!/bin/bash
# spaced JACK calls, alternating libraries, several rounds: the median call of each run (gpu_jack_p50.sh default build_ab/lib_x.so ...)
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
for round in 1 2 3 4; do
  for v in "$@"; do
    if [ "$v" = "default" ]; then r=$(python scripts/jack_loop.py ${CALLS:-3000} 500 256 2>/dev/null | tail -1); else r=$(MCCONV_LIB=$v python scripts/jack_loop.py ${CALLS:-3000} 500 256 2>/dev/null | tail -1); fi
    echo "[$v] $(echo "$r" | grep -o 'p50 [0-9.]* p90 [0-9.]*') mean $(echo "$r" | cut -d' ' -f1)"
  done
done | tee gpurun_out/jack_p50.txt
