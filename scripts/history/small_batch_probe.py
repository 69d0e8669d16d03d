#!/usr/bin/env python3
"""ms per device-buffer batch of T blocks (config 3) for a given stream_threshold: which kernel should short batches take?"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cuda_audio_amd.engine import Convolution
from cuda_audio_amd.synth import make_input, make_ir

taps = int(os.environ.get("TAPS", "441000")); nref = int(os.environ.get("NREF", "524288"))
dev = torch.device("cuda:0")
irs = [make_ir(taps, seed=5678), make_ir(taps, seed=5680)]
for thr in (int(a) for a in sys.argv[1:]):
    for T in (8, 16, 24, 32, 48, 64):
        c = Convolution("p", nref, max_batch=64, stream_threshold=thr)
        for i, ir in enumerate(irs):
            c.prepare(i, ir)
        for h in (0, 1):
            c.cc[h].value.update(select=h, vsteps=0)
        x = torch.from_numpy(make_input(T * 256)).to(dev)
        o = torch.zeros(2, T * 256, device=dev)
        for _ in range(300):
            c.process_device(x[0].data_ptr(), x[1].data_ptr(), o[0].data_ptr(), o[1].data_ptr(), T)
        c.sync()
        t0 = time.perf_counter()
        n = 400
        for _ in range(n):
            c.process_device(x[0].data_ptr(), x[1].data_ptr(), o[0].data_ptr(), o[1].data_ptr(), T)
        c.sync()
        dt = (time.perf_counter() - t0) / n
        c.enable_kernel_timing(True)
        c.process_device(x[0].data_ptr(), x[1].data_ptr(), o[0].data_ptr(), o[1].data_ptr(), T)
        ks = c.kernel_stats()
        c.close()
        print(f"stream_threshold {thr:3d}  T {T:3d}: {dt * 1e6:7.1f} us per batch ({T * 256 / 44100 / dt:8.0f} x)  resident={ks['resident']} levels={ks['fast_levels']}")
