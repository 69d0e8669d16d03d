#!/bin/bash
# bench.py's N > 1 flow end to end on ONE GPU, started the way the driver starts it (plain `python bench.py --gpus 2`: the ranks are
# its own child processes): two ranks share the card, gloo in place of RCCL (the numbers mean nothing; what counts is that all three
# layouts run and the sharded_check entries are ~1e-8)
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
timeout -k 10 900 python bench.py --gpus 2 --backend gloo --steps 4 --warmup 2 "$@" > gpurun_out/b_2rank.json 2> gpurun_out/b_2rank.err || { tail -8 gpurun_out/b_2rank.err; exit 1; }
python - <<'PY'
import json
d = json.loads(open("gpurun_out/b_2rank.json").read().strip().splitlines()[-1])
print(d["n_gpus"], d["value"], d["ms_per_step"], d["scaling"], d["config"]["blocks_per_step"], d.get("sharded_check"))
for k in ("north_star_layout", "north_star_layout_direct_mac"):
    n = d.get(k) or {}
    print(k, n.get("value"), n.get("ms_per_step"), n.get("blocks_per_step"), n.get("sum_over_partitions"), n.get("sharded_check"))
PY
