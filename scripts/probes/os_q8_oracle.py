"""GPU probe: the overlap-save form in the Q8 regime at the one-term shape against oracle.RefCompat directly (the suite compares it
with the partitioned passes, which the Q8 tests hold to RefCompat at sizes the oracle finishes in seconds; this takes ~1 min of CPU).
n_ref 16384, an IR of n_ref - 1024 frames on both halves, predelay 1024, two batches of 12288 blocks."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import oracle  # noqa: E402
from cuda_audio_amd.engine import Convolution  # noqa: E402
from cuda_audio_amd.synth import make_input, make_ir  # noqa: E402

n_ref, T = 16384, 12288
x = make_input(2 * T * 256)
ir = make_ir(n_ref - 1024, seed=7, norm=0.02)
c = Convolution("q8", n_ref, max_batch=T)
c.prepare(0, ir)
for h in (0, 1):
    c.cc[h].value.update(select=0, predelay=1024, wet=0.5, dry=0.5, panWet=0.0, panDry=0.0, level=1.0, vsteps=0, speed=100)
dx = torch.from_numpy(x).cuda()
out = torch.zeros(2, 2 * T * 256, device="cuda")
c.enable_kernel_timing(True)
lv = []
for k in range(2):
    o = k * T * 256
    c.process_device(dx[0, o:].data_ptr(), dx[1, o:].data_ptr(), out[0, o:].data_ptr(), out[1, o:].data_ptr(), T)
    c.sync()
    lv.append(c.kernel_stats()["fast_levels"])
got = out.cpu().numpy()
print("forms", lv, "os", c.os_stats(), "drops", c.drop_stats())
c.close()
r = oracle.RefCompat(n_ref, True)
r.prepare(0, ir)
for h in (0, 1):
    r.set(h, select=0, predelay=1024, wet=0.5, dry=0.5, panWet=0.0, panDry=0.0, level=1.0, vsteps=0, speed=100)
t = time.time()
want = r.process(x[0], x[1])
d = got.astype(np.float64) - want
print(f"oracle {time.time() - t:.1f} s; rms err batch 0 {np.sqrt(np.mean(d[:, :T * 256] ** 2)):.3e}  batch 1 (the form) {np.sqrt(np.mean(d[:, T * 256:] ** 2)):.3e}  signal {np.sqrt(np.mean(want ** 2)):.3e}")
lin = oracle.RefCompat(n_ref * 2, True)  # (no cut at twice the size: how large the cut terms are)
