#!/bin/bash
# block-sliced rank (rank 0 of an emulated world) under switches: gpu_emu_ab.sh default VAR=value ...
cd ${GRAFT_REPO_ROOT:-$(pwd)}
for v in "$@"; do
  ( if [ "$v" != "default" ]; then export "$v"; fi
    python bench.py --steps 10 --warmup 3 --force-sharded --emulate-world 8 --layouts blocks --no-check 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('[$v]', d['value'], d['ms_per_step'])" )
done
