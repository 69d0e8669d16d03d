#!/usr/bin/env python3
"""One-block-per-call (JACK) path only: for rocprofv3 runs and wall-clock timing."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cuda_audio_amd.engine import Convolution
from cuda_audio_amd.synth import make_input, make_ir

n = int(sys.argv[1]) if len(sys.argv) > 1 else 500
period = int(sys.argv[2]) if len(sys.argv) > 2 else 256  # JACK period: 256, 512 or 1024 frames
c = Convolution("lat", 524288, max_batch=4, period=period)
c.prepare(0, make_ir(441000, seed=5678))
c.prepare(1, make_ir(441000, seed=5680))
c.cc[1].value.select = 1
x = make_input(period)
for _ in range(int(os.environ.get("WARM", "100"))):
    c.onProcess(x[0], x[1])
t0 = time.perf_counter()
for _ in range(n):
    c.onProcess(x[0], x[1])
dt = (time.perf_counter() - t0) / n
print(f"wall_us_per_block {dt*1e6:.1f} rtf {period/44100/dt:.1f} avgRuntime_ms {c.avgRuntime():.4f}")
