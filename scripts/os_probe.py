"""GPU probe: the overlap-save form of long settled batches (csrc/ossave.hip.h) against the second-level-transform
path (MCCONV_OS=0), batch by batch, and the state it leaves for the calls that follow (a short batch, single periods).
Usage: python scripts/os_probe.py [n_ref taps batches]   (defaults 131072 88200 3)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cuda_audio_amd.engine import Convolution  # noqa: E402
from cuda_audio_amd.synth import make_input, make_ir  # noqa: E402

n_ref = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
taps = int(sys.argv[2]) if len(sys.argv) > 2 else 88200
nb = int(sys.argv[3]) if len(sys.argv) > 3 else 3
predelay = int(sys.argv[4]) if len(sys.argv) > 4 else 0
dev = torch.device("cuda:0")
irs = [make_ir(taps, seed=5678), make_ir(taps, seed=5680)]


def rms(a):
    a = np.asarray(a, np.float64)
    return float(np.sqrt(np.mean(a * a)))


def run(os_on, T=None):
    os.environ["MCCONV_OS"] = "1" if os_on else "0"
    c = Convolution(fftSize=n_ref, max_batch=65536)
    for i, ir in enumerate(irs):
        c.prepare(i, ir)
    for half in (0, 1):
        c.cc[half].value.update(select=half, predelay=predelay, wet=0.5, dry=0.5, panWet=0.1 * half, panDry=0.0, level=1.0, vsteps=0, speed=100)
    T = T or c.preferred_batch(40000)
    tail = 512
    x = make_input((nb * T + tail + 8) * 256)
    d_in = torch.from_numpy(x).to(dev)
    d_out = torch.zeros(nb, 2, T * 256, device=dev)
    lv = []
    c.enable_kernel_timing(True)
    for k in range(nb):
        o = k * T * 256
        c.process_device(d_in[0, o:].data_ptr(), d_in[1, o:].data_ptr(), d_out[k, 0].data_ptr(), d_out[k, 1].data_ptr(), T)
        c.sync()
        lv.append(c.kernel_stats()["fast_levels"])
    # what follows: a short batch (resident MAC), then single periods (JACK path)
    o = nb * T * 256
    d_t = torch.zeros(2, tail * 256, device=dev)
    c.process_device(d_in[0, o:].data_ptr(), d_in[1, o:].data_ptr(), d_t[0].data_ptr(), d_t[1].data_ptr(), tail)
    c.sync()
    per = []
    for j in range(8):
        a = (nb * T + tail + j) * 256
        per.append(np.stack(c.onProcess(x[0, a:a + 256], x[1, a:a + 256])))
    st = c.os_stats()
    out = d_out.cpu().numpy()
    t = d_t.cpu().numpy()
    c.close()
    return T, out, t, np.concatenate(per, axis=1), lv, st


T, ref, ref_t, ref_p, lv0, st0 = run(False)
print("reference path: T", T, "levels", lv0, "os", st0)
T2, got, got_t, got_p, lv1, st1 = run(True, T)
print("overlap-save  : T", T2, "levels", lv1, "os", st1)
sig = rms(ref[1:])
print("signal rms", sig)
for k in range(nb):
    d = got[k] - ref[k]
    print(f"batch {k}: rms diff {rms(d):.3e}  max {np.abs(d).max():.3e}")
    # where: per 1024-block stretch
    e = (d.astype(np.float64) ** 2).reshape(2, -1, 256).mean(axis=(0, 2)) ** 0.5
    worst = np.argsort(e)[-5:][::-1]
    print("   worst blocks", [(int(b), float(e[b])) for b in worst])
    step = max(1, len(e) // 16)
    print("   per stretch", " ".join(f"{rms(e[i:i + step]):.1e}" for i in range(0, len(e), step)))
print(f"short batch after: rms diff {rms(got_t - ref_t):.3e}  (first 8 blocks {rms(got_t[:, :2048] - ref_t[:, :2048]):.3e})")
print(f"periods after    : rms diff {rms(got_p - ref_p):.3e}")
