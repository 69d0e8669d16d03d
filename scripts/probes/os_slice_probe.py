"""GPU probe: block-sliced engines in the overlap-save form (two virtual ranks on one card) against an unsliced engine."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cuda_audio_amd.engine import Convolution  # noqa: E402
from cuda_audio_amd.sharded import slice_bounds  # noqa: E402
from cuda_audio_amd.synth import make_input, make_ir  # noqa: E402

n_ref, taps, world = 131072, (88200, 80000), 2
T, nbat, pd = 26000, 3, int(sys.argv[1]) if len(sys.argv) > 1 else 0
x = make_input(nbat * T * 256)
irs = [make_ir(t, seed=5678 + 2 * j, norm=0.02) for j, t in enumerate(taps)]
dx = torch.from_numpy(x).cuda()


def mk(os_on):
    os.environ["MCCONV_OS"] = "1" if os_on else "0"
    c = Convolution(fftSize=n_ref, max_batch=T)
    for i, ir in enumerate(irs):
        c.prepare(i, ir)
    for h in (0, 1):
        c.cc[h].value.update(select=h, predelay=pd, wet=0.7, dry=0.5, panWet=0.25 * (1 - h), panDry=0.0, level=1.0 - 0.1 * h, vsteps=0, speed=100)
    c.use_torch_stream()
    return c


whole = mk(False)
ranks = [mk(True) for _ in range(world)]
for k in range(nbat):
    xin = dx[:, k * T * 256:(k + 1) * T * 256].contiguous()
    o = torch.zeros(2, T * 256, device="cuda")
    whole.process_device(xin[0].data_ptr(), xin[1].data_ptr(), o[0].data_ptr(), o[1].data_ptr(), T)
    torch.cuda.synchronize()
    for r, c in enumerate(ranks):
        first, count = slice_bounds(T, world, r, 1)
        oo = torch.zeros(2, count * 256, device="cuda")
        print("batch", k, "rank", r, "first", first, "count", count, flush=True)
        c.process_slice_device(xin[0].data_ptr(), xin[1].data_ptr(), oo[0].data_ptr(), oo[1].data_ptr(), T, first, count)
        torch.cuda.synchronize()
        d = (oo - o[:, first * 256:(first + count) * 256]).double()
        print("   os batches", c.os_stats(), "rms diff", float((d * d).mean().sqrt()), flush=True)
