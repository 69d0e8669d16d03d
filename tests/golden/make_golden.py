#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/.

The reference holds no fixtures for this path and cannot be
built or run here (CUDA + cuFFT), so these vectors come from the independent
numpy/pocketfft restatement oracle/refcompat_np.py of conv.cu:207-253, 287-466 —
NOT from the C oracle they are used to check, and not from the reference
itself ("parity unpinned").  Each .npz holds inputs, IRs, parameters and the
expected float64 output.  Run from the repo root:  python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cuda_audio_amd.synth import make_input, make_ir  # noqa: E402
from oracle.refcompat_np import RefCompatNp  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
BASE = dict(select=0, predelay=0, wet=0.5, dry=0.5, panWet=0.0, panDry=0.0, level=1.0, vsteps=0, speed=100)


def case(name, n_ref, nb, taps, p0, p1, x=None, norm=0.05):
    x = make_input(nb * 256) if x is None else x
    irs = [make_ir(taps[0], seed=11, norm=norm), make_ir(taps[1], seed=22, norm=norm)]
    r = RefCompatNp(n_ref, three_mult=True)
    for i, ir in enumerate(irs):
        r.prepare(i, ir)
    for half, p in ((0, p0), (1, p1)):
        for k, v in p.items():
            r.cc[half][k] = np.float32(v) if isinstance(v, float) else v
    y = r.process(x[0], x[1])
    assert np.abs(y).max() < 1.0
    np.savez_compressed(os.path.join(HERE, name + ".npz"), x=x.astype(np.float32), ir0=irs[0], ir1=irs[1],
                        expected=y.astype(np.float64), n_ref=np.int64(n_ref),
                        params=np.array(json.dumps([p0, p1])))
    print(name, "rms", float(np.sqrt(np.mean(y * y))))


def main():
    nb = 24
    n = nb * 256
    case("cold_start_defaults", 4096, nb, (2500, 2800), dict(BASE), dict(BASE, select=1))
    case("unequal_params", 4096, nb, (2500, 2800),
         dict(BASE, predelay=300, wet=0.7, dry=0.3, panWet=0.25, panDry=-0.5, level=0.9),
         dict(BASE, select=1, predelay=17, wet=0.4, dry=0.6, panWet=-0.75, panDry=0.1, level=0.8))
    case("predelay_1024", 4096, nb, (2500, 2800), dict(BASE, predelay=1024), dict(BASE, select=1))
    rng = np.random.default_rng(7)
    dc = (0.2 + 0.01 * rng.standard_normal((2, n))).astype(np.float32)
    case("dc_heavy_input", 4096, nb, (2500, 2800), dict(BASE), dict(BASE, select=1), x=dc, norm=0.01)
    alt = (0.2 * (-1.0) ** np.arange(n) + 0.01 * rng.standard_normal((2, n))).astype(np.float32)
    case("alternating_input", 4096, nb, (2500, 2800), dict(BASE), dict(BASE, select=1), x=alt, norm=0.01)
    case("slow_fade", 4096, nb, (2500, 2800), dict(BASE, vsteps=20), dict(BASE, select=1, vsteps=7))
    case("nref_131072", 131072, 12, (20000, 30000), dict(BASE, predelay=1024), dict(BASE, select=1))


if __name__ == "__main__":
    main()
