"""Seeded synthetic inputs and impulse responses.

Shared by bench.py and tests/ so that the GPU engine, the oracle and the CPU
baseline all see the same samples.  The reference's own IR bank
(/root/reference/ir, Voxengo / EchoThief licences) is never read at run time.
"""
import numpy as np

FS = 44100
BLOCK = 256


def make_input(n_frames, seed=1234, channels=2, dc=0.01, amp=0.25):
    """uniform(-amp, amp) + dc per channel, seed + channel; float32 [channels, n]."""
    out = np.empty((channels, n_frames), dtype=np.float32)
    step = 1 << 24  # (drawn in pieces: the same stream as one draw, without a float64 copy of the whole channel)
    for ch in range(channels):
        rng = np.random.default_rng(seed + ch)
        for o in range(0, n_frames, step):
            n = min(step, n_frames - o)
            out[ch, o:o + n] = (rng.uniform(-amp, amp, n) + dc).astype(np.float32)
    return out


def make_ir(taps, seed=5678, norm=1.0):
    """Stereo exponentially decaying Gaussian noise, -60 dB at the end, sum h^2 = norm
    per channel; float32 [taps, 2] (interleaved L,R frames like WavFile holds)."""
    out = np.empty((taps, 2), dtype=np.float32)
    t = np.arange(taps, dtype=np.float64)
    tau = max(taps / 6.9, 1.0)
    env = np.exp(-t / tau)
    for ch in range(2):
        rng = np.random.default_rng(seed + ch)
        h = rng.standard_normal(taps) * env
        h *= np.sqrt(norm / np.sum(h * h))
        out[:, ch] = h.astype(np.float32)
    return out
