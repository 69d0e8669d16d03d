#!/bin/bash
# Round-3 bench records (profiles/r3_*.json): the headline, the reference's shipped operating point, the batch-length sweep.
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out/r3
run() { name=$1; shift; timeout -k 10 400 python bench.py "$@" > gpurun_out/r3/$name.json 2> gpurun_out/r3/$name.err || { echo "$name failed"; tail -3 gpurun_out/r3/$name.err; }; }
run bench_cfg3 --steps 200 --warmup 20
run shipped_defaults --shipped-defaults --steps 100 --warmup 10
for T in 256 1024 4096 8192 32320 129296; do
  run sweep_T$T --blocks $T --steps 100 --warmup 10 --no-cpu-baseline --no-latency --no-host-io
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r3/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "unreadable", e); continue
    r = d.get("roofline") or {}
    lm = d.get("latency_mode") or {}
    print(f.split("/")[-1], "rtf", d["value"], "ms/step", d["ms_per_step"], "kernel", r.get("kernel"), r.get("kernel_avg_ms"), "frac", r.get("frac"),
          "parity", (d.get("parity") or {}).get("rms_err"), "jack us", lm.get("us_per_block_wall"), lm.get("us_per_call_period_spaced"))
PY
