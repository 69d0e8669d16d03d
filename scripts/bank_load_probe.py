#!/usr/bin/env python3
"""Load a bank of 152 IRs - the size of the reference's ir/all.index (main.cu:72-80), synthetic taps with the shipped bank's
spread of lengths (759 ... 352 193 frames) - into one engine at N_ref = 524288 and report the load time, the
device memory it keeps and what one long batch adds (second-level spectra of the two sounding IRs only).  Round 3: the
fast-FIR components are built on first use; run under rocprofv3 --kernel-trace to see that no
k_polyphase launch appears."""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from cuda_audio_amd.engine import Convolution  # noqa: E402
from cuda_audio_amd.synth import make_input, make_ir  # noqa: E402

n_ref, nir = 524288, 152
rng = np.random.default_rng(7)
lengths = np.exp(rng.uniform(np.log(759), np.log(352193), nir)).astype(int)
lengths[0], lengths[-1] = 759, 352193
torch.cuda.init()
free0, total = torch.cuda.mem_get_info()
c = Convolution("bank", n_ref, max_batch=8192)
free1, _ = torch.cuda.mem_get_info()
t0 = time.perf_counter()
for i, n in enumerate(lengths):
    c.prepare(i, make_ir(int(n), seed=100 + i, norm=0.05))
c.sync()
t_load = time.perf_counter() - t0
free2, _ = torch.cuda.mem_get_info()
c.cc[0].value.update(select=3)
c.cc[1].value.update(select=77)
T = 8192
x = make_input(T * 256)
dx = torch.from_numpy(x).cuda()
out = torch.zeros(2, T * 256, device="cuda")
c.use_torch_stream()
for _ in range(3):
    c.process_device(dx[0].data_ptr(), dx[1].data_ptr(), out[0].data_ptr(), out[1].data_ptr(), T)
c.sync()
free3, _ = torch.cuda.mem_get_info()
pstride = c.debug_dims()["pstride"]
print(json.dumps({
    "irs": nir, "n_ref": n_ref, "frames_min_max": [int(lengths.min()), int(lengths.max())],
    "load_seconds": round(t_load, 3), "load_ms_per_ir": round(t_load / nir * 1e3, 2),
    "engine_MB_before_any_ir": round((free0 - free1) / 1e6, 1),
    "bank_resident_MB": round((free1 - free2) / 1e6, 1), "bank_MB_per_ir": round((free1 - free2) / 1e6 / nir, 2),
    "spectra_MB_per_ir_by_construction": round(256 * pstride * 16 / 1e6, 2),
    "added_by_three_long_batches_MB": round((free2 - free3) / 1e6, 1),
    "note": "round 2 built the fast-FIR components eagerly: + 59.8 MB and three k_polyphase launches per IR (9.1 GB for this bank); "
            "now they exist only after a batch that selects the fast-FIR MAC (MCCONV_FFT2=0)",
}))
c.close()
