#!/usr/bin/env python3
"""Two config-3 engines on their own streams, the second one started half a step late: do a step's streaming launches
(k_fwd, k_inv_wet) run in the gaps the second-level transform of the OTHER engine leaves at the HBM side?
usage: stagger_probe.py [delay_blocks] [steps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cuda_audio_amd.engine import Convolution  # noqa: E402
from cuda_audio_amd.synth import make_input, make_ir  # noqa: E402

delay = int(sys.argv[1]) if len(sys.argv) > 1 else 0
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
T = 129296
dev = torch.device("cuda", 0)
eng = []
for p in range(2):
    e = Convolution(f"s{p}", 524288, max_batch=T, device=0)
    e.prepare(0, make_ir(441000, seed=5678 + 4 * p))
    e.prepare(1, make_ir(441000, seed=5680 + 4 * p))
    for h in (0, 1):
        e.cc[h].value.update(select=h, predelay=0, dry=0.5, wet=0.5, panDry=0.0, panWet=0.0, level=1.0, vsteps=0)
    eng.append(e)
x = torch.from_numpy(make_input(T * 256)).to(dev)
out = [torch.zeros(2, T * 256, device=dev) for _ in range(2)]


def run(n, which=(0, 1)):
    for _ in range(n):
        for p in which:
            eng[p].process_device(x[0].data_ptr(), x[1].data_ptr(), out[p][0].data_ptr(), out[p][1].data_ptr(), T)


run(3)
for e in eng:
    e.sync()
for name, which in (("one engine", (0,)), ("two engines", (0, 1))):
    if delay and len(which) == 2:
        eng[1].process_device(x[0].data_ptr(), x[1].data_ptr(), out[1][0].data_ptr(), out[1][1].data_ptr(), delay)
    t0 = time.perf_counter()
    run(steps, which)
    for e in eng:
        e.sync()
    dt = (time.perf_counter() - t0) / steps
    print(f"{name}: {dt * 1e3:.4f} ms per round of {len(which)} step(s) = {dt * 1e3 / len(which):.4f} ms per step (delay {delay} blocks)")
for e in eng:
    e.close()
