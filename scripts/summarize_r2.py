#!/usr/bin/env python3
"""Condense gpurun_out/prof_r2_<tag>/ (scripts/profile_r2.sh) into profiles/:
   r2_<tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats table of the bench command
   r2_<tag>_summary.md         per-kernel time, HBM-side bytes, SQ counters and what they say
   r2_hbm_traffic.json         {kernel: {hbm_bytes_per_launch, ...}} read (and labelled) by bench.py
   r2_g2_counters.json         SQ counters of the dominant kernel and the derived busy fractions
usage: summarize_r2.py <tag> <dominant kernel substring> [blocks_per_launch]"""
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CUS, SIMDS = 256, 4


def counters(src, name):
    files = glob.glob(os.path.join(src, name, "*", "*_counter_collection.csv"))
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    if not files:
        return out, dur
    seen = set()
    for r in csv.DictReader(open(max(files, key=os.path.getmtime))):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        out[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        did = r.get("Dispatch_Id")
        if (k, did) not in seen:
            seen.add((k, did))
            dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    return out, dur


def head():
    try:
        return subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], stderr=subprocess.DEVNULL).decode().strip()
    except Exception:
        return None


def mean(v):
    return sum(v) / len(v) if v else None


def main(tag, dominant, blocks=None):
    src = os.path.join(ROOT, "gpurun_out", f"prof_r2_{tag}")
    dst = os.path.join(ROOT, "profiles")
    ks = max(glob.glob(os.path.join(src, "stats", "*", "*_kernel_stats.csv")), key=os.path.getmtime)
    shutil.copy(ks, os.path.join(dst, f"r2_{tag}_kernel_stats.csv"))
    rows = list(csv.DictReader(open(ks)))
    cmd = open(os.path.join(src, "command.txt")).read().strip()
    allc = {}
    durs = {}
    for p in ("pmc_fetch", "pmc_write", "pmc_sq1", "pmc_sq2", "pmc_sq3", "pmc_tcc"):
        c, d = counters(src, p)
        for k, cs in c.items():
            allc.setdefault(k, {}).update({n: mean(v) for n, v in cs.items()})
        for k, v in d.items():
            durs.setdefault(k, {})[p] = mean(v) / 1e3
    lines = [f"# rocprofv3 summary - round 2, `{tag}`", "", f"Command: `rocprofv3 --kernel-trace --stats --output-format csv -- python3 {cmd.replace('--prewarm-ms 0', '--prewarm-ms 100')}` (durations: 100 ms of untimed steps first, clocks warm as in the bench itself);",
             "counters from separate `--kernel-trace --pmc ...` runs of the same command with `--prewarm-ms 0` (scripts/profile_r2.sh), means per launch.", "",
             "| kernel | calls | avg us | % of GPU time | FETCH_SIZE KiB (raw) | WRITE_SIZE KiB | HBM-side MB (2 x fetch + write) |", "|---|---|---|---|---|---|---|"]
    traffic = {}
    for r in rows:
        k = r["Name"].split("(")[0].replace("void ", "")
        c = allc.get(k, {})
        f, w = c.get("FETCH_SIZE"), c.get("WRITE_SIZE")
        hb = int((2 * f + w) * 1024) if f is not None and w is not None else None
        if hb is not None:
            traffic[k] = {"hbm_bytes_per_launch": hb, "fetch_size_kib_raw": round(f, 1), "write_size_kib": round(w, 1),
                          "avg_us": round(float(r["AverageNs"]) / 1e3, 2), "calls": int(r["Calls"])}
        lines.append(f"| {k[:50]} | {r['Calls']} | {float(r['AverageNs']) / 1e3:.2f} | {float(r['Percentage']):.2f} | "
                     f"{'' if f is None else round(f, 1)} | {'' if w is None else round(w, 1)} | {'' if hb is None else round(hb / 1e6, 1)} |")
    lines += ["", "FETCH_SIZE is doubled (gfx950 reports half the bytes of wide coalesced reads, MI355X_MICROARCH.md, HBM); WRITE_SIZE is exact",
              "for 16-byte-per-lane stores. Infinity-Cache hits are counted: these are bytes at the L2's memory side, an upper bound of HBM traffic.", ""]
    dk = [k for k in allc if dominant in k]
    g2 = None
    if dk:
        k = dk[0]
        c = allc[k]
        t_us = None
        for r in rows:
            if dominant in r["Name"]:
                t_us = float(r["AverageNs"]) / 1e3
        g = lambda n: c.get(n)
        lines += [f"## `{k}`: what the waves do", "", "| counter | mean per launch |", "|---|---|"]
        for n in sorted(c):
            lines.append(f"| {n} | {c[n]:.6g} |")
        der = {}
        wc = g("SQ_WAVE_CYCLES")
        if wc:
            for n, lab in (("SQ_WAIT_ANY", "waiting (s_waitcnt / barrier)"), ("SQ_WAIT_INST_ANY", "issue-stalled"), ("SQ_ACTIVE_INST_ANY", "issuing"),
                           ("SQ_ACTIVE_INST_VALU", "issuing VALU"), ("SQ_ACTIVE_INST_LDS", "issuing LDS"), ("SQ_ACTIVE_INST_VMEM", "issuing VMEM"),
                           ("SQ_ACTIVE_INST_SCA", "issuing scalar")):
                if g(n) is not None:
                    der[f"wave_time_{lab}"] = round(g(n) / wc, 4)
        gui = g("GRBM_GUI_ACTIVE")
        if gui and t_us:
            der["clock_GHz"] = round(gui / 8 / (t_us * 1e3), 3)  # sum over 8 XCDs (guide: reads high on short dispatches)
        if gui:
            cyc = gui / 8  # shader cycles of the launch
            if g("SQ_LDS_IDX_ACTIVE") is not None:
                der["lds_busy_frac_of_cu_cycles"] = round(g("SQ_LDS_IDX_ACTIVE") / (cyc * CUS), 4)
                der["lds_bank_conflict_frac_of_lds_cycles"] = round((g("SQ_LDS_BANK_CONFLICT") or 0) / g("SQ_LDS_IDX_ACTIVE"), 4)
            if g("SQ_ACTIVE_INST_VALU") is not None:
                # SQ_ACTIVE_INST_* count quad-cycles (4 shader cycles) per SIMD
                der["valu_busy_frac_of_simd_cycles"] = round(4 * g("SQ_ACTIVE_INST_VALU") / (cyc * CUS * SIMDS), 4)
        if g("SQ_INSTS_VALU") and g("SQ_WAVES"):
            der["valu_insts_per_wave"] = round(g("SQ_INSTS_VALU") / g("SQ_WAVES"), 1)
            der["lds_insts_per_wave"] = round((g("SQ_INSTS_LDS") or 0) / g("SQ_WAVES"), 1)
            der["vmem_rd_insts_per_wave"] = round((g("SQ_INSTS_VMEM_RD") or 0) / g("SQ_WAVES"), 1)
        if g("TCC_HIT_sum") is not None and g("TCC_MISS_sum") is not None:
            der["l2_hit_rate"] = round(g("TCC_HIT_sum") / max(g("TCC_HIT_sum") + g("TCC_MISS_sum"), 1), 4)
        der = {a: b for a, b in der.items() if b is not None}
        lines += ["", "Derived:", ""] + [f"- {a}: {b}" for a, b in der.items()]
        der["profiled_at_commit"] = head()
        g2 = {"kernel": k, "avg_us": t_us, "counters": c, "derived": der, "command": cmd, "profiled_at_commit": head(),
              "durations_us_per_pass": durs.get(k)}
        if blocks:
            g2["blocks_per_launch"] = int(blocks)
        json.dump(g2, open(os.path.join(dst, f"r2_{tag}_counters.json"), "w"), indent=1)
    bench = ""
    for line in open(os.path.join(src, "stats.log"), errors="replace"):
        if line.startswith('{"metric"'):
            bench = line.strip()
    if bench:
        lines += ["", "bench.py line of the kernel-trace run:", "", "```", bench, "```"]
    open(os.path.join(dst, f"r2_{tag}_summary.md"), "w").write("\n".join(lines) + "\n")
    tj = os.path.join(dst, "r2_hbm_traffic.json")
    data = json.load(open(tj)) if os.path.exists(tj) else {}
    for k, v in traffic.items():
        if blocks:
            v["blocks_per_launch"] = int(blocks)
        v["command"] = cmd
        v["profiled_at_commit"] = head()
        data.setdefault(tag, {})[k] = v
    json.dump(data, open(tj, "w"), indent=1)
    print("\n".join(lines[:60]))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else None)
