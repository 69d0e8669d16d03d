#!/usr/bin/env python3
"""Condense gpurun_out/prof_r4_<tag>/ (scripts/profile_r4.sh) into profiles/:
   r4_<tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats table of the bench command
   r4_<tag>_summary.md         per-kernel time, HBM-side bytes, SQ counters and what they say
   r4_hbm_traffic.json         {kernel: {hbm_bytes_per_launch, ...}} read (and labelled) by bench.py
   r4_<tag>_counters.json         SQ counters of the dominant kernel and the derived busy fractions
usage: summarize_r4.py <tag> <dominant kernel substring> [blocks_per_launch]"""
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
    src = os.path.join(ROOT, "gpurun_out", f"prof_r4_{tag}")
    dst = os.path.join(ROOT, "profiles")
    ks = max(glob.glob(os.path.join(src, "stats", "*", "*_kernel_stats.csv")), key=os.path.getmtime)
    shutil.copy(ks, os.path.join(dst, f"r4_{tag}_kernel_stats.csv"))
    rows = list(csv.DictReader(open(ks)))
    cmd = open(os.path.join(src, "command.txt")).read().strip()
    allc = {}
    durs = {}
    for p in ("pmc_fetch", "pmc_write", "pmc_sq1", "pmc_sq2", "pmc_sq3", "pmc_tcc", "pmc_mem1", "pmc_mem2", "pmc_mem3", "pmc_mem4"):
        c, d = counters(src, p)
        for k, cs in c.items():
            allc.setdefault(k, {}).update({n: mean(v) for n, v in cs.items()})
        for k, v in d.items():
            durs.setdefault(k, {})[p] = mean(v) / 1e3
    lines = [f"# rocprofv3 summary - round 4, `{tag}`", "", f"Command: `rocprofv3 --kernel-trace --stats --output-format csv -- python3 {cmd.replace('--prewarm-ms 0', '--prewarm-ms 100')}` (durations: 100 ms of untimed steps first, clocks warm as in the bench itself);",
             "counters from separate `--kernel-trace --pmc ...` runs of the same command with `--prewarm-ms 0` (scripts/profile_r4.sh), means per launch.", "",
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
    dk = [k for k in allc if k == dominant] or [k for k in allc if dominant in k]
    g2 = None
    if dk:
        k = dk[0]
        c = allc[k]
        t_us = None
        for r in rows:
            if r["Name"].split("(")[0].replace("void ", "") == k:
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
        # the memory path of a CU (round 3): L1 -> L2 requests (64-byte requests on gfx950: stated, not calibrated), their mean
        # round trip, the share of the launch during which an L1 could not accept another request, address-unit occupancy
        if g("TCP_TCC_READ_REQ_sum") and t_us:
            rd, wr = g("TCP_TCC_READ_REQ_sum"), g("TCP_TCC_WRITE_REQ_sum") or 0.0
            # request sizes: 128-byte reads (16.0 M requests for the 2.01 GB this kernel reads through L1: windows 671 MB with their
            # overlaps + four paths' spectra per item 1342 MB), 64-byte writes (8.7 M for the 530 MB of sums; TCC_EA0_WRREQ_64B agrees)
            der["l1_to_l2_read_requests"] = rd
            der["l1_to_l2_read_MB_at_128B_per_request"] = round(rd * 128 / 1e6, 1)
            der["l1_to_l2_read_GBps_per_cu"] = round(rd * 128 / (t_us * 1e-6) / 1e9 / CUS, 2)
            der["l1_to_l2_write_GBps_per_cu"] = round(wr * 64 / (t_us * 1e-6) / 1e9 / CUS, 2)
            if g("TCP_TCC_READ_REQ_LATENCY_sum"):
                der["l1_read_request_mean_latency_cycles"] = round(g("TCP_TCC_READ_REQ_LATENCY_sum") / rd, 1)
                if gui:
                    # Little: requests in flight per CU = request rate x latency
                    der["l1_read_requests_in_flight_per_cu"] = round(g("TCP_TCC_READ_REQ_LATENCY_sum") / (gui / 8) / CUS, 1)
        if g("TCP_PENDING_STALL_CYCLES_sum") is not None and gui:
            der["l1_pending_stall_frac_of_cu_cycles"] = round(g("TCP_PENDING_STALL_CYCLES_sum") / (gui / 8 * CUS), 4)
        if g("TCP_TCP_TA_DATA_STALL_CYCLES_sum") is not None and gui:
            der["l1_to_ta_data_stall_frac_of_cu_cycles"] = round(g("TCP_TCP_TA_DATA_STALL_CYCLES_sum") / (gui / 8 * CUS), 4)
        if g("TA_BUSY_avr") is not None and gui:
            der["ta_busy_frac_of_cycles"] = round(g("TA_BUSY_avr") / (gui / 8), 4)  # (TA_BUSY_avr: busy cycles, mean over the address units)
        if g("TCC_REQ_sum") is not None and t_us:
            der["l2_requests"] = g("TCC_REQ_sum")
            if g("TCC_EA0_RDREQ_sum") is not None:
                der["l2_memory_side_read_requests"] = g("TCC_EA0_RDREQ_sum")
                der["l2_memory_side_read_requests_to_dram"] = g("TCC_EA0_RDREQ_DRAM_sum")
            if g("TCC_EA0_RDREQ_LEVEL_sum") and g("TCC_EA0_RDREQ_sum"):
                der["l2_memory_side_read_mean_latency_cycles"] = round(g("TCC_EA0_RDREQ_LEVEL_sum") / g("TCC_EA0_RDREQ_sum"), 1)
        der = {a: b for a, b in der.items() if b is not None}
        lines += ["", "Derived:", ""] + [f"- {a}: {b}" for a, b in der.items()]
        der["profiled_at_commit"] = head()
        g2 = {"kernel": k, "avg_us": t_us, "counters": c, "derived": der, "command": cmd, "profiled_at_commit": head(),
              "durations_us_per_pass": durs.get(k)}
        if blocks:
            g2["blocks_per_launch"] = int(blocks)
        json.dump(g2, open(os.path.join(dst, f"r4_{tag}_counters.json"), "w"), indent=1)
    bench = ""
    for line in open(os.path.join(src, "stats.log"), errors="replace"):
        if line.startswith('{"metric"'):
            bench = line.strip()
    if bench:
        lines += ["", "bench.py line of the kernel-trace run:", "", "```", bench, "```"]
    open(os.path.join(dst, f"r4_{tag}_summary.md"), "w").write("\n".join(lines) + "\n")
    tj = os.path.join(dst, "r4_hbm_traffic.json")
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
