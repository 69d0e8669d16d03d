#!/usr/bin/env python3
"""does a large max_batch (ring / scratch strides) slow the 32320-block step down?  tcap_probe.py 32768 262144"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cuda_audio_amd.engine import Convolution
from cuda_audio_amd.synth import make_input, make_ir

dev = torch.device("cuda:0")
irs = [make_ir(441000, seed=5678), make_ir(441000, seed=5680)]
T = 32320
x = torch.from_numpy(make_input(T * 256)).to(dev)
o = torch.zeros(2, T * 256, device=dev)
for mb in (int(a) for a in sys.argv[1:]):
    c = Convolution("p", 524288, max_batch=mb)
    for i, ir in enumerate(irs):
        c.prepare(i, ir)
    for h in (0, 1):
        c.cc[h].value.update(select=h, vsteps=0)
    for _ in range(30):
        c.process_device(x[0].data_ptr(), x[1].data_ptr(), o[0].data_ptr(), o[1].data_ptr(), T)
    c.sync()
    c.enable_kernel_timing(True)
    c.kernel_stats(reset=True)
    t0 = time.perf_counter()
    n = 100
    for _ in range(n):
        c.process_device(x[0].data_ptr(), x[1].data_ptr(), o[0].data_ptr(), o[1].data_ptr(), T)
    c.sync()
    dt = (time.perf_counter() - t0) / n
    ks = c.kernel_stats()
    c.close()
    print(f"max_batch {mb:7d}: {dt * 1e3:.4f} ms per step, k_g2_mac {ks['total_ms'] / ks['launches']:.4f} ms")
