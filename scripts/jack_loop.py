#!/usr/bin/env python3
"""The JACK path in a loop (one 256-frame period per mc_process call, config 3; JACK_SHIPPED=1: the reference's shipped operating point) - the program scripts/profile_jack.sh
runs under rocprofv3.  Prints microseconds per call."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cuda_audio_amd.engine import Convolution  # noqa: E402
from cuda_audio_amd.synth import make_input, make_ir  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
gap_us = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0  # idle time between calls (a JACK period is 5805 us)
F = int(sys.argv[3]) if len(sys.argv) > 3 else 256         # frames per period (256, 512, 1024)
if os.environ.get("JACK_SHIPPED") == "1":  # the reference's shipped operating point (settings.txt:19,38-45): the Q8 regime
    c = Convolution("jack", 131072, max_batch=8, device=0, period=F)
    c.prepare(0, make_ir(131072 - 1024, seed=5678))
    for h in (0, 1):
        c.cc[h].value.update(select=0, predelay=1024, vsteps=0)
else:
    c = Convolution("jack", 524288, max_batch=8, device=0, period=F)
    c.prepare(0, make_ir(441000, seed=5678))
    c.prepare(1, make_ir(441000, seed=5680))
    for h in (0, 1):
        c.cc[h].value.update(select=h, vsteps=0)
x = make_input(F)
fp = C.POINTER(C.c_float)
bufs = [np.ascontiguousarray(x[0]), np.ascontiguousarray(x[1]), np.zeros(F, np.float32), np.zeros(F, np.float32)]
p = [b.ctypes.data_as(fp) for b in bufs]
for _ in range(300):
    c._L.mc_process(c._h, p[0], p[1], p[2], p[3], F)
if gap_us <= 0:
    t0 = time.perf_counter()
    for _ in range(n):
        c._L.mc_process(c._h, p[0], p[1], p[2], p[3], F)
    dt = (time.perf_counter() - t0) / n
    print(f"{dt * 1e6:.2f} us per {F}-frame call, {F / 44100 / dt:.1f} x real time, avgRuntime {c.avgRuntime() * 1e3:.2f} us")
else:
    # what a JACK client sees: the host is idle between periods, the call's own duration is what counts
    tot = 0.0
    worst = 0.0
    all_d = []
    for _ in range(n):
        t1 = time.perf_counter()
        while (time.perf_counter() - t1) * 1e6 < gap_us:
            pass
        t1 = time.perf_counter()
        c._L.mc_process(c._h, p[0], p[1], p[2], p[3], F)
        d = time.perf_counter() - t1
        tot += d
        worst = max(worst, d)
        all_d.append(d)
    q = np.percentile(np.array(all_d) * 1e6, [50, 90, 99, 99.9])
    print(f"{tot / n * 1e6:.2f} us per {F}-frame call ({F / 44100 / (tot / n):.0f} x) with {gap_us:.0f} us idle between calls (worst {worst * 1e6:.1f} us; "
          f"p50 {q[0]:.2f} p90 {q[1]:.2f} p99 {q[2]:.2f} p99.9 {q[3]:.2f}), avgRuntime {c.avgRuntime() * 1e3:.2f} us")
c.close()
