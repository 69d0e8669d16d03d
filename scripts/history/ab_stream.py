#!/usr/bin/env python3
"""Interleaved in-process A/B of streaming-MAC launch shapes (workgroup size, chunk count)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cuda_audio_amd.engine import Convolution
from cuda_audio_amd.synth import make_input, make_ir

cfgs = [tuple(map(int, a.split(","))) for a in sys.argv[1:]] or [(256, 2), (512, 1), (256, 1), (512, 2)]
ir_a, ir_b = make_ir(441000, seed=5678), make_ir(441000, seed=5680)
x = make_input(2048 * 256)
engines = []
for nt, nc in cfgs:
    os.environ["MCCONV_STREAM_NT"], os.environ["MCCONV_NCHUNK"] = str(nt), str(nc)
    c = Convolution("ab", 524288, max_batch=256)
    c.prepare(0, ir_a); c.prepare(1, ir_b); c.cc[1].value.select = 1
    c.process(x[0], x[1])  # fill the delay line (2048 blocks > P)
    engines.append(c)
res = {cfg: [] for cfg in cfgs}
for rnd in range(6):
    for cfg, c in zip(cfgs, engines):
        c.enable_kernel_timing(True); c.kernel_stats(reset=True)
        for b in range(200):
            c.onProcess(x[0, b*256:(b+1)*256], x[1, b*256:(b+1)*256])
        ks = c.kernel_stats(); c.enable_kernel_timing(False)
        res[cfg].append(ks["total_ms"] / ks["launches"] * 1e3)
for cfg in cfgs:
    v = sorted(res[cfg]); print(f"NT={cfg[0]} nchunk={cfg[1]}  median {v[len(v)//2]:.2f} us  min {v[0]:.2f}  max {v[-1]:.2f}   avgRuntime {engines[cfgs.index(cfg)].avgRuntime()*1e3:.1f} us")
