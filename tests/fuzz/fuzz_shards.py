#!/usr/bin/env python3
"""Randomised runs of the two multi-GPU layouts as virtual ranks on ONE card, against oracle.RefCompat (GPU box; not part of the test suite).
Every run picks a layout - partition shards finished on rank 0 (`reduce`), partition shards with every rank finishing its run of
blocks (`reduce_scatter`), or output blocks sliced across ranks with no exchange (`slices`) - a world of 2..4 ranks, IR lengths up
to n_ref - 1024, a period of 256 / 512 frames, batches of random length, and controller traffic between batches (select, predelay,
wet, speed, pans, level).  usage: fuzz_shards.py [first_seed] [runs] [os]
`os` (round 4, lab build: MCCONV_LIB=build_ab/lib_lab.so): the overlap-save form from one block on (MCCONV_OS_MIN=1), so that the block slices of
every settled batch run as a segment that reads its history from the batch in front of the slice."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle as oracle_mod  # noqa: E402
from cuda_audio_amd.engine import Convolution  # noqa: E402
from cuda_audio_amd.sharded import partitions_for, shard_bounds, slice_bounds  # noqa: E402
from cuda_audio_amd.synth import make_input  # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 1
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 20
if len(sys.argv) > 3 and sys.argv[3] == "os":
    os.environ["MCCONV_OS_MIN"] = "1"
os_batches = 0
TOL = 1e-5
cmap = (21, 22, 23, 24, 25, 26, 27, 28)
arr = (C.c_uint8 * 8)(*cmap)
bad = 0
count = {}
for seed in range(first, first + runs):
    rng = np.random.default_rng(seed)
    layout = str(rng.choice(["reduce", "reduce_scatter", "slices"]))
    world = int(rng.integers(2, 5))
    n_ref = int(rng.choice([4096, 8192, 16384]))
    period = int(rng.choice([256, 256, 512]))
    pm = period // 256
    nirs = int(rng.integers(2, 4))  # (a sharded engine refuses more IRs cross-fading than it has voices: no merge across shards)
    full = n_ref - 1024
    lens = [full if rng.random() < 0.4 else int(rng.integers(300, full + 1)) for _ in range(nirs)]
    irs = []
    for L in lens:
        h = rng.standard_normal((L, 2)) * np.exp(-np.arange(L) / (1.5 * L))[:, None]
        irs.append((h * np.sqrt(0.003 / L)).astype(np.float32))
    tmax = 48 * pm
    ncalls = (5 * n_ref // 256 + 160) // pm
    x = make_input(ncalls * period, seed=100 + seed)
    ref = oracle_mod.RefCompat(n_ref, True)
    P = max(partitions_for(L, n_ref) for L in lens)
    eng = []
    for r in range(world):
        kw = {}
        if layout != "slices":
            pb, pe = shard_bounds(P, world, r)
            kw = dict(part_begin=pb, part_end=pe)
        e = Convolution(f"r{r}", n_ref, max_batch=tmax, stream_threshold=8, period=period, **kw)
        e.use_torch_stream()
        eng.append(e)
    for i, ir in enumerate(irs):
        ref.prepare(i, ir)
        for e in eng:
            e.prepare(i, ir)

    def cc(half, ctl, val):
        oracle_mod.handle_cc(ref.cc(half), cmap, ctl, val, ref.num_irs())
        for e in eng:
            assert e._L.mc_handle_cc(e._h, half, arr, ctl, val) == 0

    for half in (0, 1):
        cc(half, 22, 16 if rng.random() < 0.5 else int(rng.integers(0, 128)))
        cc(half, 25, int(rng.integers(0, 3)))  # short cross-fades: at most two IRs sounding per half most of the time
    dx = torch.from_numpy(x).cuda()
    got = torch.zeros(2, ncalls * period, device="cuda")
    want = np.zeros((2, ncalls * period))
    q, last_select = 0, -100
    n_fixed = int(rng.integers(1, 13)) * world  # block slices: a rank's slice starts at the same block of every batch
    try:
        while q < ncalls:
            if q and rng.random() < 0.5:
                half, ctl, val = int(rng.integers(0, 2)), int(rng.choice(cmap)), int(rng.integers(0, 128))
                if ctl == 22 and rng.random() < 0.5:
                    val = 16
                if ctl == 25:
                    val = int(rng.integers(0, 3))
                if ctl == 28:
                    val = int(rng.integers(64, 128))
                if layout == "slices" and ctl == 22:
                    ctl = 24  # (a block-sliced engine refuses a predelay change: wet instead)
                if ctl != 21 or q - last_select > 40:  # (selects far enough apart that the voices never overflow)
                    cc(half, ctl, val)
                    if ctl == 21:
                        last_select = q
            n = int(min(rng.integers(world, 49), ncalls - q))
            if layout == "slices":
                n = n_fixed
                if q + n > ncalls:
                    break
            if layout != "reduce" and n < world:
                n = min(world, ncalls - q)
            T = n * pm
            s = slice(q * period, (q + n) * period)
            want[:, s] = ref.process(x[0, s], x[1, s], block=period)
            xin = dx[:, s].contiguous()
            if layout == "slices":
                if n < world:
                    break
                for r, e in enumerate(eng):
                    f, c = slice_bounds(T, world, r, pm)
                    o = torch.zeros(2, c * 256, device="cuda")
                    e.process_slice_device(xin[0].data_ptr(), xin[1].data_ptr(), o[0].data_ptr(), o[1].data_ptr(), T, f, c)
                    got[:, q * period + f * 256:q * period + (f + c) * 256] = o
            else:
                parts = [torch.zeros(2 * T * 256, device="cuda") for _ in eng]
                for e, p in zip(eng, parts):
                    e.partial_device(xin[0].data_ptr(), xin[1].data_ptr(), p.data_ptr(), T)
                total = torch.stack(parts).sum(0)
                if layout == "reduce" or n < world:
                    o = torch.zeros(2, T * 256, device="cuda")
                    eng[0].finish_device(xin[0].data_ptr(), xin[1].data_ptr(), total.data_ptr(), o[0].data_ptr(), o[1].data_ptr(), T)
                    for e in eng[1:]:
                        e.finish_device(None, None, None, None, None, T)
                    got[:, s] = o
                else:
                    tot2 = total.view(2, T * 256)
                    for r, e in enumerate(eng):
                        f, c = slice_bounds(T, world, r, pm)
                        sl = tot2[:, f * 256:(f + c) * 256].contiguous()
                        o = torch.zeros(2, c * 256, device="cuda")
                        e.finish_slice_device(xin[0].data_ptr(), xin[1].data_ptr(), sl.data_ptr(), o[0].data_ptr(), o[1].data_ptr(), T, f, c)
                        got[:, q * period + f * 256:q * period + (f + c) * 256] = o
            q += n
        note = ""
    except Exception as ex:  # an engine may refuse a call (e.g. more IRs cross-fading than a shard has voices): reported; what ran before it is compared
        note = f" [stopped at call {q}: {str(ex)[:120]}]"
    torch.cuda.synchronize()
    err = float(np.sqrt(np.mean((got.cpu().numpy()[:, :q * period] - want[:, :q * period]) ** 2))) if q else 0.0
    for e in eng:
        os_batches += e.os_stats()["batches"]
        e.close()
    count[layout] = count.get(layout, 0) + 1
    flag = "" if err <= TOL else "   <-- FAIL"
    bad += err > TOL
    print(f"seed {seed}: {layout} x{world}, n_ref {n_ref}, period {period}, IRs {lens}, rms {err:.3e} (signal {np.sqrt(np.mean(want[:, :max(q, 1) * period] ** 2)):.3e}){note}{flag}", flush=True)
print(f"{runs} runs {count}, {bad} above {TOL}; rank-batches in the overlap-save form: {os_batches}")
sys.exit(1 if bad else 0)
