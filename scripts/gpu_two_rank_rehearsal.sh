#!/bin/bash
# bench.py's N > 1 flow end to end on ONE GPU: two ranks share the card, gloo in place of RCCL (the numbers mean nothing;
# what counts is that both layouts run and both sharded_check entries are ~1e-8)
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
timeout -k 10 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 \
  bench.py --gpus 2 --backend gloo --steps 4 --warmup 2 > gpurun_out/b_2rank.json 2> gpurun_out/b_2rank.err || { tail -5 gpurun_out/b_2rank.err; exit 1; }
python - <<'PY'
import json
d = json.loads(open("gpurun_out/b_2rank.json").read().strip().splitlines()[-1])
print(d["n_gpus"], d["value"], d["ms_per_step"], d["scaling"], d.get("sharded_check"), (d.get("north_star_layout") or {}).get("sharded_check"))
PY
