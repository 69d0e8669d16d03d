#!/bin/bash
# Latency distribution of the 256-frame JACK path with the host idle between periods: 20000 calls per variant, alternating.
cd ${GRAFT_REPO_ROOT:-$(pwd)}
for round in 1 2; do
  for v in default MCCONV_TAGGED_IO=0; do
    ( if [ "$v" != "default" ]; then export "$v"; fi
      echo "[$v] $(python scripts/jack_loop.py 20000 200 256 2>/dev/null | tail -1)" )
  done
done
