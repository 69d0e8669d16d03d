#!/bin/bash
# JACK path under switches: back to back and with 500 us idle between calls.  gpu_jack_ab.sh default VAR=value ...
cd ${GRAFT_REPO_ROOT:-$(pwd)}
for round in 1 2; do
for v in "$@"; do
  ( if [ "$v" != "default" ]; then export "$v"; fi
    for F in ${PERIODS:-256}; do
      echo "[$v] $(python scripts/jack_loop.py 4000 0 $F 2>/dev/null | tail -1) | spaced: $(python scripts/jack_loop.py 1500 500 $F 2>/dev/null | tail -1)"
    done )
done
done
