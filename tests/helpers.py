"""Shared helpers for the parity tests (oracle vs HIP engine)."""
import numpy as np

# tolerance stated by BASELINE.json's north_star: 1e-5 RMS (float32 engine vs float64 oracle)
RMS_TOL = 1e-5


def rms(a):
    a = np.asarray(a, dtype=np.float64)
    return float(np.sqrt(np.mean(a * a)))


def apply_params(target, p0, p1, is_oracle):
    """Set both halves' CC values on an oracle engine or a cuda_audio_amd.Convolution."""
    for half, p in ((0, p0), (1, p1)):
        if is_oracle:
            target.set(half, **p)
        else:
            target.cc[half].value.update(**p)


BASE = dict(select=0, predelay=0, wet=0.5, dry=0.5, panWet=0.0, panDry=0.0, level=1.0, vsteps=0, speed=100)
