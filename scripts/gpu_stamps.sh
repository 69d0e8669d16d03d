#!/bin/bash
# In-kernel stamps of k_g2_mac / k_g2_duo (diagnostic builds -DG2_STAMPS=1): where a workgroup's cycles go.
#   gpu_stamps.sh "<lib> [VAR=value ...]" ...     one short bench run per argument, the kernel's printf lines kept
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
for v in "$@"; do
  echo "== $v"
  set -- $v
  lib=$1; shift
  ( for kv in "$@"; do export "$kv"; done
    MCCONV_LIB=$PWD/$lib timeout -k 10 120 python bench.py --steps 2 --warmup 1 --prewarm-ms 0 --no-cpu-baseline --no-latency --no-host-io --no-parity > gpurun_out/stamps_tmp.log 2>&1
    echo "rc $?" )
  grep -a "g2 wg" gpurun_out/stamps_tmp.log | tail -12
  grep -a -o '"kernel_avg_ms": [0-9.]*' gpurun_out/stamps_tmp.log
done
