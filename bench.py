#!/usr/bin/env python3
"""Headline benchmark: real-time factor of the convolution hot path.

Workload (BASELINE.json configs[2], the one the metric is quoted on): stereo
44.1 kHz, 256-frame blocks, 10 s / 441 000-tap IR (P = 1723 partitions,
N_ref = 524288), reference routing (2 inputs x 2 outputs = 4 convolution
paths), fp32.  One "step" = one batch of --blocks (default: the engine's preferred length, 32320) consecutive blocks pushed
through forward FFT -> partition x bin MAC -> inverse FFT -> overlap-add ->
predelay / Q1-Q2 terms / clamp / dry mix, inputs and outputs resident in HBM.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

N > 1, default (--shard blocks): every GPU holds the whole IR set and is fed the
same batch of N x --blocks blocks; rank r finishes output blocks [r, r + 1) x
--blocks of it - independent units, no data-path collective, weak scaling.
--shard partitions: IR partitions sharded over the ranks, the partial wet blocks
summed with an RCCL reduce (the layout that also shortens one real-time period).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FS = 44100
BLOCK = 256
HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: 8.0 TB/s spec
FP32_PEAK_TFLOPS = 157.3  # vector = f32-MFMA rate


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--blocks", type=int, default=0,
                    help="blocks per step (batch length T). 0 (default): what the engine prefers up to 32768 blocks "
                         "(mc_preferred_batch: whole chunks of the second-level transform minus one halo block) - 32320 = 187.6 s "
                         "of audio = five chunks of 8192 - 1728 + 1 blocks for the 1723-partition IR; 8 block-sliced ranks "
                         "then stay within mc_config.max_batch (262144)")
    ap.add_argument("--taps", type=int, default=441000)
    ap.add_argument("--fft-size", type=int, default=524288, help="reference fftSize (N_ref)")
    ap.add_argument("--mode", choices=["resident", "stream"], default="resident",
                    help="resident: IR held on chip across the batch; stream: every block re-reads IR+delay line")
    ap.add_argument("--precision", choices=["fp32", "fp16"], default="fp32",
                    help="fp16: IR spectra and delay line stored as half for the streaming sweep (implies --mode stream)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--no-latency", action="store_true", help="skip the 1-block-per-call (JACK) measurement")
    ap.add_argument("--shard", choices=["blocks", "partitions"], default="blocks",
                    help="N > 1. blocks: every GPU is fed the same batch and finishes its slice of the output blocks - no "
                         "data-path collective, the slices are gathered to rank 0 (default: batch throughput). partitions: "
                         "every GPU holds 1/N of the IR partitions and the partial wet blocks are summed with RCCL (the "
                         "layout that also shortens a single real-time period)")
    ap.add_argument("--exchange", choices=["gather", "none"], default="none",
                    help="--shard blocks: leave every rank's finished slice on the GPU that computed it (default: like the "
                         "inputs, the outputs of the batch path live in HBM; at ~150 GB/s of finished audio per GPU any "
                         "funnel into one GPU is bound by its xGMI links, not by the convolution) or gather the slices "
                         "on rank 0 over RCCL, overlapped with the next batch")
    ap.add_argument("--collective", choices=["reduce", "allreduce"], default="reduce",
                    help="--shard partitions: sum of partial wet blocks to rank 0 (default) or to every rank")
    ap.add_argument("--no-overlap", action="store_true", help="N > 1: finish each batch before starting the next")
    ap.add_argument("--pipeline", action="store_true",
                    help="mc_config.pipeline: run the post stage of batch k on a second stream under the MAC of batch k + 1 "
                         "(same output bits; a few percent either way depending on the batch length - off by default, so that the "
                         "dominant kernel is timed alone, as rocprofv3 sees it)")
    ap.add_argument("--no-check", action="store_true",
                    help="N > 1: skip the untimed comparison of the sharded pipeline with an unsharded engine on rank 0")
    ap.add_argument("--emulate-world", type=int, default=0,
                    help="with --force-sharded on one GPU: use the partition shard rank 0 of this many ranks would own")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="N > 1 collective backend. gloo is a rehearsal of the launch path on a box with fewer GPUs than "
                         "ranks (ranks share cards, the sum goes through host memory): never a headline number")
    ap.add_argument("--force-sharded", action="store_true",
                    help="use the partial / collective / finish path even with one rank (rehearsal on one GPU)")
    return ap.parse_args()


def cpu_threads():
    """Threads for the CPU baseline: the cgroup CPU quota of this box when there is one (a GPU box shows all
    host CPUs but is throttled to its share), else what OpenMP reports."""
    import oracle

    n = oracle.max_threads()
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    return n


def cpu_baseline(ir, ir_b, x, seconds):
    """oracle Cpu32 (float32 OpenMP partitioned overlap-save) on a bounded sample of the same workload:
    chunks of 128 blocks until `seconds` of wall clock have passed."""
    import oracle

    threads = cpu_threads()
    eng = oracle.Cpu32(ir, ir_b)
    g = np.array([0.5, 0.5, 0.5, 0.5], np.float32)
    chunk, avail = 128, x.shape[1] // BLOCK
    eng.process(x[0, : chunk * BLOCK], x[1, : chunk * BLOCK], g, g, threads)  # warm-up, untimed
    t0 = time.perf_counter()
    done = 0
    while time.perf_counter() - t0 < seconds:
        o = (done % max(avail - chunk, 1))
        eng.process(x[0, o * BLOCK : (o + chunk) * BLOCK], x[1, o * BLOCK : (o + chunk) * BLOCK], g, g, threads)
        done += chunk
    dt = time.perf_counter() - t0
    eng.close()
    return {
        "value": round(done * BLOCK / FS / dt, 3),
        "unit": "x realtime",
        "cores": threads,
        "kind": "port",
        "sample": f"{done} blocks ({done * BLOCK / FS:.1f} s of audio) of the same stereo/{ir.shape[0]}-tap workload, "
                  f"oracle/oracle.c orc_cpu32 (own radix-2 FFT, OpenMP over bins), {dt:.1f} s wall",
    }


def main():
    a = parse()
    # The contract is ONE JSON line on stdout.  Libraries below us write there too (gloo's connection banner, RCCL
    # at some debug levels): from here on everything that goes to file descriptor 1 lands on stderr, and the JSON
    # line is written to the original stdout at the end.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    import torch
    import torch.distributed as dist

    from cuda_audio_amd.engine import Convolution
    from cuda_audio_amd.synth import make_input, make_ir

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world:
        if world == 1 and a.gpus > 1:
            raise SystemExit("launch with torch.distributed.run for --gpus > 1")
    if a.backend == "gloo":
        local %= max(torch.cuda.device_count(), 1)  # rehearsal: ranks may share a card
        a.collective = "allreduce"                  # gloo has no reduce on device tensors
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    sharded = world > 1 or a.force_sharded
    if sharded:
        # RCCL prints its version banner to stdout at NCCL_DEBUG=VERSION; the contract is ONE JSON line on stdout
        if not os.environ.get("MCCONV_KEEP_NCCL_DEBUG"):
            os.environ.pop("NCCL_DEBUG", None)
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        elif a.backend == "gloo":
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    ir = make_ir(a.taps, seed=5678)
    if a.blocks <= 0:
        probe = Convolution("probe", a.fft_size, max_batch=32768, device=local)
        probe.prepare(0, ir)
        a.blocks = probe.preferred_batch(32768)
        probe.close()
    T = a.blocks
    P = (min(a.taps, a.fft_size - 1024) + BLOCK - 1) // BLOCK
    # shard bounds: multiples of 16 partitions
    from cuda_audio_amd.sharded import shard_bounds

    from cuda_audio_amd.sharded import slice_bounds

    shard_world = world if world > 1 else (a.emulate_world if a.force_sharded and a.emulate_world > 1 else 1)
    by_blocks = sharded and a.shard == "blocks"
    if by_blocks:
        # weak scaling: every rank finishes --blocks output blocks of a batch of N x --blocks
        T = a.blocks * shard_world
    first, count = slice_bounds(T, shard_world, rank) if by_blocks else (0, T)
    if shard_world > 1 and not by_blocks:
        pb, pe = shard_bounds(P, shard_world, rank)
    else:
        pb, pe = 0, 0
    if a.precision == "fp16":
        a.mode = "stream"
    thr = (T + 1) if a.mode == "stream" else 0
    # pipelined batches (the post stage of batch k under the MAC of batch k + 1) wherever no cross-GPU sum sits
    # between the two halves of a batch
    pipelined = a.pipeline and (not sharded or by_blocks) and a.precision == "fp32"
    eng = Convolution("bench", a.fft_size, max_batch=T, device=local, part_begin=pb,
                      part_end=pe if (shard_world > 1 and not by_blocks) else 0, stream_threshold=min(thr, 16385), precision=a.precision,
                      pipeline=pipelined)
    if shard_world > 1 and not by_blocks and pe == pb:
        raise SystemExit("empty shard; use fewer ranks")
    # two distinct IRs (seed 5678 + path): in1 -> (L,R) through IR 0, in2 -> (L,R) through IR 1,
    # i.e. four different convolution paths, so the 4-path byte count has no shared spectra
    ir_b = make_ir(a.taps, seed=5680)
    eng.prepare(0, ir)
    eng.prepare(1, ir_b)
    for h in (0, 1):
        eng.cc[h].value.update(select=h, predelay=0, dry=0.5, wet=0.5, panDry=0.0, panWet=0.0, level=1.0, vsteps=0)

    n_distinct = 4  # rotate through a few distinct input batches
    xs = make_input(n_distinct * T * BLOCK, seed=1234)
    d_in = torch.from_numpy(xs).to(dev)
    d_out = torch.zeros(2, T * BLOCK, device=dev)
    d_parts = [torch.zeros(2 * T * BLOCK, device=dev) for _ in range(2)] if (sharded and not by_blocks) else None
    # block slices: this rank's output blocks of a batch (double-buffered), gathered on rank 0
    d_slices = [torch.zeros(2, count * BLOCK, device=dev) for _ in range(3)] if by_blocks else None
    d_gather = ([[torch.zeros(2, count * BLOCK, device=dev) for _ in range(world)] for _ in range(2)]
                if by_blocks and rank == 0 and shard_world == world and a.exchange == "gather" else None)
    # one compute stream for the engine, the torch ops and (as the stream the collectives order themselves
    # against) RCCL: partial -> reduce -> finish are then ordered by the streams, not by host synchronisation
    torch.cuda.synchronize()
    comp = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(comp)
    eng.use_torch_stream(comp)
    root_only = a.collective == "reduce"
    pending = []

    kept = None  # check mode: rank 0 keeps a copy of every finished batch

    def retire():
        # second half of the oldest batch in flight: wait for its collective, then predelay / clamp / dry on the sum
        work, part, i1, i2 = pending.pop(0)
        work.wait()  # makes the compute stream wait for the collective; the host does not block
        if rank == 0 or not root_only:
            eng.finish_device(i1, i2, part.data_ptr(), d_out[0].data_ptr(), d_out[1].data_ptr(), T)
            if kept is not None and rank == 0:
                kept.append(d_out.clone())
        else:
            eng.finish_device(None, None, None, None, None, T)

    class _Done:
        def wait(self):
            pass

    def gather_slices(sl, k):
        # the only exchange of the block-sliced layout: count * 2 KB per rank and batch to rank 0, off the data path
        if shard_world != world or a.exchange == "none":  # (one emulated rank of several: nothing to gather)
            return _Done()
        if a.backend == "gloo":  # rehearsal: through host memory
            h = sl.cpu()
            hl = [torch.zeros_like(h) for _ in range(world)] if rank == 0 else None
            dist.gather(h, hl, dst=0)
            if rank == 0:
                for g, t in zip(d_gather[k % 2], hl):
                    g.copy_(t)
            return _Done()
        return dist.gather(sl, d_gather[k % 2] if rank == 0 else None, dst=0, async_op=True)

    def retire_slices():
        work, k = pending.pop(0)
        work.wait()
        if kept is not None and rank == 0:
            kept.append(torch.cat(d_gather[k % 2], dim=1) if d_gather is not None else d_slices[k % 3].clone())

    ungathered = []  # batches whose slices are computed (or in the pipeline) but not handed to the gather yet

    def hand_over(older_only):
        # the engine's stream waits for the post stage of the batches to be gathered - for all of them at the end,
        # for all but the newest during the run (so that the next batch's MAC is not queued behind that post stage)
        if pipelined:
            eng.fence_older() if older_only else eng.fence()
        keep = ungathered[-1:] if older_only else []
        for j in ungathered[:len(ungathered) - len(keep)]:
            while len(pending) >= 2:  # the gather buffers are two deep
                retire_slices()
            pending.append((gather_slices(d_slices[j % 3], j), j))
        ungathered[:] = keep

    def step(k):
        o = (k % n_distinct) * T * BLOCK
        i1, i2 = d_in[0, o:].data_ptr(), d_in[1, o:].data_ptr()
        if not sharded:
            eng.process_device(i1, i2, d_out[0].data_ptr(), d_out[1].data_ptr(), T)
            return
        if by_blocks:
            # at most two gathers in flight; the slice buffer of batch k - 3 is free once its gather has been waited for
            while len(pending) > (0 if a.no_overlap else 1):
                retire_slices()
            sl = d_slices[k % 3]
            eng.process_slice_device(i1, i2, sl[0].data_ptr(), sl[1].data_ptr(), T, first, count)
            ungathered.append(k)
            hand_over(older_only=pipelined and not a.no_overlap)
            if a.no_overlap:
                while pending:
                    retire_slices()
            return
        part = d_parts[k % 2]
        eng.partial_device(i1, i2, part.data_ptr(), T)
        if root_only:
            work = dist.reduce(part, dst=0, async_op=True)
        else:
            work = dist.all_reduce(part, async_op=True)
        # the reduce of batch k overlaps the MAC of batch k+1: batch k-1 is finished now
        if pending and not a.no_overlap:
            retire()
        pending.append((work, part, i1, i2))
        if a.no_overlap:
            retire()

    def drain():
        if by_blocks:
            hand_over(older_only=False)
        elif pipelined:
            eng.fence()
        while pending:
            retire_slices() if by_blocks else retire()

    # settle the cold-start cross-fade (Q7) so the timed region is steady state
    for k in range(max(a.warmup, 1)):
        step(k)
    drain()
    torch.cuda.synchronize()
    eng.enable_kernel_timing(True)
    eng.kernel_stats(reset=True)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(a.steps):
        step(k)
    drain()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ks = eng.kernel_stats()
    eng.enable_kernel_timing(False)
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    blocks = a.steps * T
    rtf = blocks * BLOCK / FS / dt
    alg_bytes = eng.algorithmic_bytes_per_block()  # this rank's share
    kern_s = ks["total_ms"] * 1e-3
    kern_avg_ms = ks["total_ms"] / max(ks["launches"], 1)
    achieved_gbs = alg_bytes * ks["blocks"] / kern_s / 1e9 if kern_s > 0 else 0.0
    flops_per_block = 8.0 * 4 * ks["partitions"] * 256  # complex MAC = 8 flop, 4 paths
    achieved_tf = flops_per_block * ks["blocks"] / kern_s / 1e12 if kern_s > 0 else 0.0

    # latency mode (what JACK sees): one 256-frame period per mc_process call, host buffers in and out,
    # the call returns when the output is on the host.  Called through ctypes with preallocated buffers
    # so that the harness adds microseconds, not the tens of microseconds of numpy allocations.
    latency = None
    if rank == 0 and not sharded and not a.no_latency:
        import ctypes as C

        eng.set_stream(None)
        L = eng._L
        fp = C.POINTER(C.c_float)
        bufs = [np.ascontiguousarray(xs[0, :BLOCK]), np.ascontiguousarray(xs[1, :BLOCK]),
                np.zeros(BLOCK, np.float32), np.zeros(BLOCK, np.float32)]
        ptrs = [b.ctypes.data_as(fp) for b in bufs]
        for _ in range(200):
            L.mc_process(eng._h, ptrs[0], ptrs[1], ptrs[2], ptrs[3], BLOCK)
        eng.enable_kernel_timing(True)
        eng.kernel_stats(reset=True)
        n_lat = 2000
        t1 = time.perf_counter()
        for _ in range(n_lat):
            L.mc_process(eng._h, ptrs[0], ptrs[1], ptrs[2], ptrs[3], BLOCK)
        lat = (time.perf_counter() - t1) / n_lat
        ks1 = eng.kernel_stats()
        eng.enable_kernel_timing(False)
        k_ms = ks1["total_ms"] / max(ks1["launches"], 1)
        # rocprofv3 duration of the same kernel in the same path, from the committed profile of this round
        prof_us = None
        try:
            import csv

            rows = list(csv.DictReader(open(os.path.join(ROOT, "profiles", "r1_jack_kernel_stats.csv"))))
            tot = sum(float(r["TotalDurationNs"]) for r in rows if "k_mac_stream" in r["Name"])
            cnt = sum(int(r["Calls"]) for r in rows if "k_mac_stream" in r["Name"])
            prof_us = tot / cnt / 1e3 if cnt else None
        except Exception:
            prof_us = None
        latency = {
            "us_per_block_wall": round(lat * 1e6, 2),
            "rtf": round(BLOCK / FS / lat, 1),
            "avg_runtime_ms": round(eng.avgRuntime(), 5),
            "mac_kernel": "k_mac_stream (every block re-reads 4 IR paths + 2 delay-line inputs)",
            "mac_kernel_us_event_bracketed": round(k_ms * 1e3, 2),
            "mac_achieved_GBps": round(alg_bytes / (k_ms * 1e-3) / 1e9, 1) if k_ms > 0 else None,
            "mac_frac_of_hbm_peak": round(alg_bytes / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if k_ms > 0 else None,
            "mac_kernel_us_rocprofv3": round(prof_us, 2) if prof_us else None,
            "mac_achieved_GBps_rocprofv3": round(alg_bytes / (prof_us * 1e-6) / 1e9, 1) if prof_us else None,
            "mac_frac_of_hbm_peak_rocprofv3": round(alg_bytes / (prof_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4) if prof_us else None,
            "note": "event-bracketed single launches include ~3 us of event overhead; the rocprofv3 figure is the "
                    "kernel's average duration in profiles/r1_jack_kernel_stats.csv (same path, same IRs)",
        }
        # the periods the reference's run scripts start jackd with (512 / 1024 frames per call)
        longer = {}
        for period in (512, 1024):
            if T % (period // BLOCK):
                continue
            eng.set_period(period)
            pb_ = [np.ascontiguousarray(xs[0, :period]), np.ascontiguousarray(xs[1, :period]),
                   np.zeros(period, np.float32), np.zeros(period, np.float32)]
            pp_ = [b.ctypes.data_as(fp) for b in pb_]
            for _ in range(200):
                L.mc_process(eng._h, pp_[0], pp_[1], pp_[2], pp_[3], period)
            t1 = time.perf_counter()
            for _ in range(1000):
                L.mc_process(eng._h, pp_[0], pp_[1], pp_[2], pp_[3], period)
            lp = (time.perf_counter() - t1) / 1000
            longer[str(period)] = {"us_per_call_wall": round(lp * 1e6, 2), "rtf": round(period / FS / lp, 1)}
        eng.set_period(BLOCK)
        latency["longer_periods"] = longer

    # Untimed: the sharded pipeline exactly as timed above (two batches in flight, collective on RCCL's stream)
    # against an unsharded engine fed the same batches from the same cold state, on rank 0.
    sharded_check = None
    if sharded and not a.no_check:
        eng.reset()
        kept = []
        nchk = 3
        for k in range(nchk):
            step(k)
        drain()
        torch.cuda.synchronize()
        if rank == 0:
            full = Convolution("check", a.fft_size, max_batch=T, device=local, stream_threshold=min(thr, 16385),
                               precision=a.precision)
            full.prepare(0, ir)
            full.prepare(1, ir_b)
            for h in (0, 1):
                full.cc[h].value.update(select=h, predelay=0, dry=0.5, wet=0.5, panDry=0.0, panWet=0.0, level=1.0, vsteps=0)
            full.use_torch_stream(comp)
            ref_out = torch.zeros_like(d_out)
            num = den = 0.0
            for k in range(nchk):
                o = (k % n_distinct) * T * BLOCK
                full.process_device(d_in[0, o:].data_ptr(), d_in[1, o:].data_ptr(), ref_out[0].data_ptr(), ref_out[1].data_ptr(), T)
                torch.cuda.synchronize()
                want = ref_out if kept[k].shape == ref_out.shape else ref_out[:, first * BLOCK:(first + count) * BLOCK]
                num += float(((kept[k] - want).double() ** 2).sum())
                den += float((want.double() ** 2).sum())
            n_el = nchk * kept[0].numel()
            sharded_check = {"batches": nchk, "rms_err_vs_unsharded": (num / n_el) ** 0.5, "rms_signal": (den / n_el) ** 0.5}
            full.close()
        kept = None
        if world > 1:
            dist.barrier()

    cpu = None
    if rank == 0 and not sharded and not a.no_cpu_baseline:
        cpu = cpu_baseline(ir, ir_b, xs, a.cpu_seconds)

    traffic = None
    tj = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if os.path.exists(tj):
        try:
            traffic = json.load(open(tj)).get(a.mode)
        except Exception:
            traffic = None

    hbm_equiv = {
        "achieved": round(achieved_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved_gbs / HBM_PEAK_GBS, 4),
        "algorithmic_bytes_per_block": alg_bytes,
        "note": "SURVEY 8(d) accounting: every block re-reads 4 IR paths + 2 delay-line inputs; bytes x blocks / MAC kernel time",
    }
    common = {"kernel": "k_mac_resident" if ks["resident"] else "k_mac_stream", "traffic": traffic,
              "kernel_avg_ms": round(kern_avg_ms, 5), "kernel_launches": ks["launches"], "blocks_per_launch": (ks["blocks"] // max(ks["launches"], 1)) if ks["launches"] else T,
              "flops_per_block": int(flops_per_block)}
    if ks["resident"] and int(ks.get("fast_levels", 0)) in (254, 255):
        # Second-level transform (k_f2_fwd + k_f2_prod): per (bin, chunk of blocks) one circular convolution of 16384 points.  Its
        # own compulsory traffic per launch: the delay-line window of both inputs, the IRs' second-level spectra (four
        # paths), the partition sums it writes; it is bound by memory (HBM / L2), not by arithmetic.
        fused = int(ks["fast_levels"]) == 254
        F2 = 8192 if fused else 16384
        Pq = int(ks["partitions"])
        blk = ks["blocks"] // max(ks["launches"], 1)
        nchunk = -(-blk // (F2 - Pq + 1))
        if fused:
            # one kernel, both inputs' spectra side by side in LDS: the window once (16 B per slot), 4 paths of
            # second-level spectra per chunk, the partition sums written
            own_bytes = 256 * (16 * (blk + nchunk * (Pq - 1)) + 4 * 8 * F2 * nchunk + 16 * blk)
        else:
            # windows of 2 inputs + their transforms parked once and read by both channel passes + 4 paths of
            # second-level spectra + the partition sums written
            own_bytes = 256 * (2 * 8 * (blk + nchunk * (Pq - 1)) + (2 + 4) * 8 * F2 * nchunk + 4 * 8 * F2 * nchunk + 16 * blk)
        own_gbs = own_bytes / (kern_avg_ms * 1e-3) / 1e9 if kern_avg_ms > 0 else 0.0
        common["kernel"] = "k_g2_mac" if fused else "k_f2_fwd + k_f2_prod"
        roofline = dict({"bound": "hbm", "achieved": round(own_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(own_gbs / HBM_PEAK_GBS, 4)}, **common)
        roofline["algorithmic_bytes_per_launch"] = own_bytes
        roofline["replaces"] = {"kernel": "partition x bin MAC (direct form)", "tflops_equivalent": round(achieved_tf, 2),
                                "frac_of_fp32_peak": round(achieved_tf / FP32_PEAK_TFLOPS, 4), "hbm_equivalent": hbm_equiv}
        roofline["note"] = ("The batch's partition sums are computed by a second-level transform along the block axis instead of "
                            "the partition x bin MAC: per bin one circular convolution per chunk of blocks (8192 points with both "
                            "inputs' spectra side by side in 133 KB of LDS, k_g2_mac; 16384 points through a stash for long IRs "
                            "and per-slot gains, k_f2_fwd + k_f2_prod) against the IRs' transformed partition sequences. "
                            "achieved = the kernel's own algorithmic bytes (delay-line window, second-level spectra of 4 paths "
                            "per chunk, partition sums written; the split form also its stash) / kernel time (HIP events on the launch stream); `traffic` "
                            "(HBM side, PMC counters) is lower because the chunks of a bin run together on one XCD and share its spectra and "
                            "the overlap of their windows in that L2; `replaces` prices the same launch as "
                            "the direct-form MAC it stands for (SURVEY 8(d) accounting). MCCONV_FFT2=0 runs the MAC kernel "
                            "(fast-FIR form), MCCONV_FFA_LEVELS=0 its direct form.")
    elif ks["resident"]:
        # the batch kernel keeps the IR on chip across the blocks of a launch (HBM traffic << algorithmic bytes):
        # its binding resource is fp32 FMA issue.  The f32 MFMA peak of gfx950 equals the vector rate (157.3 TF).
        roofline = dict({"bound": "mfma", "achieved": round(achieved_tf, 2), "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved_tf / FP32_PEAK_TFLOPS, 4)}, **common)
        lv = int(ks.get("fast_levels", 0))
        executed_tf = achieved_tf * (0.75 ** lv)
        roofline["executed"] = {"fast_fir_levels": lv, "multiply_add_fraction": round(0.75 ** lv, 4),
                                "tflops": round(executed_tf, 2), "frac": round(executed_tf / FP32_PEAK_TFLOPS, 4)}
        roofline["hbm_equivalent"] = hbm_equiv
        roofline["note"] = ("fp32 complex MAC on the vector ALU (v_pk_fma_f32; no MFMA instruction is used - the f32 MFMA rate "
                            "of gfx950 equals the vector rate, so the peak is the same 157.3 TFLOP/s). achieved = ALGORITHMIC flops "
                            "of the partition x bin MAC, 8 flop x 4 paths x partitions x 256 bins x blocks / MAC kernel time (HIP "
                            "events on the launch stream). The kernel runs the convolution along the block axis in fast-FIR form "
                            "(polyphase components, `executed.fast_fir_levels` nested levels): it issues (3/4)^levels of those "
                            "multiply-adds, so `achieved` can exceed the peak; `executed` is what the ALUs do. "
                            "traffic = HBM bytes per launch from FETCH_SIZE/WRITE_SIZE (profiles/).")
    else:
        roofline = dict({"bound": "hbm", "achieved": round(achieved_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved_gbs / HBM_PEAK_GBS, 4)}, **common)
        roofline["algorithmic_bytes_per_block"] = alg_bytes
        roofline["fp32_tflops"] = round(achieved_tf, 2)
        roofline["note"] = ("streaming MAC: every block re-reads IR spectra and delay line; achieved = algorithmic bytes x blocks / "
                            "kernel time. With many blocks per launch the 21 MB working set is served from L2/MALL, so achieved can "
                            "exceed the HBM peak; one block per launch (latency_mode) is the HBM/MALL-bound case.")
    if rank == 0:
        line = {
            "metric": "real-time factor (frames/s / 44.1k), stereo block=256, 10 s IR",
            "value": round(rtf, 2),
            "unit": "x realtime",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak" if (by_blocks or (not sharded and a.shard == "blocks")) else "strong",
            "vs_baseline": None,
            "dtype": "f32" if a.precision == "fp32" else "f16 storage, f32 accumulate",
            "data": "synthetic",
            "config": {
                "workload": f"stereo 44.1 kHz, 256-frame blocks, {a.taps}-tap IR ({P} partitions, N_ref {a.fft_size}), "
                            f"2x2 path matrix, {T} blocks per step, "
                            + ("IR spectra re-read for every block (streaming MAC)" if a.mode == "stream" else
                               "IR resident across the batch (sum over partitions: " + str(roofline.get("kernel")) + ")"),
                "blocks_per_step": T,
                "partitions": P,
                "paths": 4,
                "mode": a.mode,
                "parallelism": "single GPU" if not sharded else
                (f"output blocks of every batch sliced over {world} GPU(s) ({count} blocks each, every GPU holds the whole IR "
                 f"and transforms the whole input); no data-path collective, "
                 + (f"slices gathered to rank 0 ({'RCCL' if a.backend == 'nccl' else 'gloo rehearsal'})" if a.exchange == "gather"
                    else "slices left on their ranks") if by_blocks else
                 f"IR partitions sharded over {world} GPU(s) + {'RCCL' if a.backend == 'nccl' else 'gloo (rehearsal)'} {a.collective} of partial wet blocks")
                + ("" if a.no_overlap else ", overlapped with the next batch"),
                "shard": (a.shard if sharded else None),
            },
            "roofline": roofline,
            "cpu_baseline": cpu,
            "latency_mode": latency,
        }
        if sharded_check is not None:
            line["sharded_check"] = sharded_check
        os.write(real_stdout, (json.dumps(line) + "\n").encode())
    eng.close()
    if sharded:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
