"""debug: output finished by k_inv_wet<true> vs by k_post"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cuda_audio_amd.engine import Convolution
from cuda_audio_amd.synth import make_input, make_ir

dev = torch.device("cuda:0")
n_ref = 131072
irs = [make_ir(88200, seed=11, norm=0.08), make_ir(70000, seed=13, norm=0.08)]
T = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
pre = int(sys.argv[2]) if len(sys.argv) > 2 else 0
x = make_input(3 * T * 256)
outs = []
for fuse in ("1", "0"):
    os.environ["MCCONV_FUSE_OUT"] = fuse
    os.environ["MCCONV_CORR_RIDE"] = os.environ.get("RIDE", fuse)
    c = Convolution("t", n_ref, max_batch=T, stream_threshold=8)
    for i, ir in enumerate(irs):
        c.prepare(i, ir)
    c.cc[0].value.update(select=0, predelay=pre, vsteps=0)
    c.cc[1].value.update(select=1, predelay=pre, vsteps=0)
    d_in = torch.from_numpy(x).to(dev)
    d_out = torch.full((3, 2, T * 256), float("nan"), device=dev)
    for k in range(3):
        o = k * T * 256
        c.process_device(d_in[0, o:].data_ptr(), d_in[1, o:].data_ptr(), d_out[k, 0].data_ptr(), d_out[k, 1].data_ptr(), T)
        c.sync()
    outs.append(d_out.cpu().numpy())
    c.close()
a, b = outs
for k in range(3):
    d = a[k] - b[k]
    bad = ~(np.abs(d) <= 1e-6)
    print("batch", k, "nan", np.isnan(a[k]).sum(), "bad", bad.sum(), "of", bad.size, "max", np.nanmax(np.abs(d)))
    if bad.any():
        idx = np.argwhere(bad)
        print("  first bad (ch, frame):", idx[:5].tolist(), " last:", idx[-3:].tolist())
        blocks = np.unique(idx[:, 1] // 256)
        print("  bad blocks:", blocks[:20], "... n =", len(blocks))
        within = np.unique(idx[:, 1] % 256)
        print("  frames within block:", within[:40], "n =", len(within))
        i = idx[0]
        print("  a", a[k][i[0], i[1]:i[1] + 8], "\n  b", b[k][i[0], i[1]:i[1] + 8])
