#!/usr/bin/env python3
"""Condense a gpurun_out/prof_<tag>_<mode>/ directory (rocprofv3 --kernel-trace
--stats pass + two --pmc passes) into profiles/<tag>_<mode>_{kernel_stats.csv,summary.md}
and update profiles/hbm_traffic.json (read by bench.py for roofline.traffic)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main(tag, mode, dominant):
    src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}_{mode}")
    dst = os.path.join(ROOT, "profiles")
    os.makedirs(dst, exist_ok=True)
    newest = lambda pat: max(glob.glob(pat), key=os.path.getmtime)
    ks = newest(os.path.join(src, "stats", "*", "*_kernel_stats.csv"))
    shutil.copy(ks, os.path.join(dst, f"{tag}_{mode}_kernel_stats.csv"))
    rows = list(csv.DictReader(open(ks)))
    pmc = {}
    for name, ctr in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
        files = glob.glob(os.path.join(src, name, "*", "*_counter_collection.csv"))
        if not files:
            continue
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(max(files, key=os.path.getmtime))):
            if r["Counter_Name"] == ctr:
                agg[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
        pmc[ctr] = {k: sum(v) / len(v) for k, v in agg.items()}
    bench = ""
    for line in open(os.path.join(src, "stats.log"), errors="replace"):
        if line.startswith('{"metric"'):
            bench = line.strip()
    lines = [f"# rocprofv3 summary — {tag}, {mode} MAC kernel", "",
             "Command: `rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 10 --warmup 2 "
             f"--mode {mode} --no-cpu-baseline --no-latency` (separate `--pmc FETCH_SIZE` and `--pmc WRITE_SIZE` passes).", "",
             "| kernel | calls | avg µs | % of GPU time | FETCH_SIZE (KiB/launch, raw) | WRITE_SIZE (KiB/launch) |", "|---|---|---|---|---|---|"]
    for r in rows:
        k = r["Name"].split("(")[0]
        f = pmc.get("FETCH_SIZE", {}).get(k)
        w = pmc.get("WRITE_SIZE", {}).get(k)
        lines.append(f"| {k[:60]} | {r['Calls']} | {float(r['AverageNs']) / 1e3:.2f} | {float(r['Percentage']):.2f} | "
                     f"{'' if f is None else round(f, 1)} | {'' if w is None else round(w, 1)} |")
    def pick(d):
        # "a+b": the dominant work is split over several kernels; their traffic adds up
        tot, found = 0.0, False
        for name in dominant.split("+"):
            for k, v in d.items():
                if name in k:
                    tot += v
                    found = True
                    break
        return tot if found else None

    f = pick(pmc.get("FETCH_SIZE", {}))
    w = pick(pmc.get("WRITE_SIZE", {}))
    traffic = None
    if f is not None and w is not None:
        # MI355X_MICROARCH.md §HBM: FETCH_SIZE counts KiB and reports 1/2 of the bytes of wide coalesced
        # reads on gfx950 -> doubled; WRITE_SIZE is exact for 16-B/lane stores.
        traffic = int((2 * f + w) * 1024)
        lines += ["", f"HBM-side traffic of `{dominant}` per launch: (2 x {f:.1f} + {w:.1f}) KiB = {traffic / 1e6:.2f} MB "
                  "(FETCH_SIZE doubled per the gfx950 correction of MI355X_MICROARCH.md §HBM; an upper bound here because "
                  "part of the reads are scalar loads, for which the counter is uncalibrated)."]
    if bench:
        lines += ["", "bench.py line of the profiled (kernel-trace) run:", "", "```", bench, "```"]
    open(os.path.join(dst, f"{tag}_{mode}_summary.md"), "w").write("\n".join(lines) + "\n")
    tj = os.path.join(dst, "hbm_traffic.json")
    data = json.load(open(tj)) if os.path.exists(tj) else {}
    data[mode] = traffic
    json.dump(data, open(tj, "w"), indent=1)
    print("\n".join(lines))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3])
