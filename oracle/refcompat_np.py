"""Second, independent restatement of the reference's per-block algorithm in
numpy (pocketfft), used to cross-check oracle.c and to generate the golden
fixtures under tests/golden/ (see tests/golden/make_golden.py).

TEST INFRASTRUCTURE ONLY; PARITY UNPINNED (oracle/oracle.h).

Follows /root/reference/src/conv.cu: prepare :207-253, onProcess :287-466,
f_interpolate :15-32, f_unpackC22R :47-73, f_pointwiseMultiplyAndScale
:102-123, f_pointwiseAdd :89-100, f_addDryInterleaved :126-140.
"""
import numpy as np

MAX_PREDELAY = 8192  # conv.h:26-28


def pan_l(p):  # conv.cu:386
    return 1 - p if p >= 0 else 1.0


def pan_r(p):  # conv.cu:387
    return 1 + p if p <= 0 else 1.0


def unpack(z):
    """f_unpackC22R incl. the DC shortcut (Q1) and unwritten Nyquist (Q2)."""
    n = len(z)
    L = np.zeros(n, dtype=np.complex128)
    R = np.zeros(n, dtype=np.complex128)
    s = np.arange(1, n // 2)
    va, vb = z[s], np.conj(z[n - s])
    la = 0.5 * (va + vb)
    lb = 1j * (-0.5 * (va - vb))
    L[s], R[s] = la, lb
    L[n - s], R[n - s] = np.conj(la), np.conj(lb)
    L[0] = z[0]  # vb = va  ->  0.5 * (va + va)
    R[0] = 0.0
    return L, R


class RefCompatNp:
    def __init__(self, fft_size, three_mult=True):
        self.N = N = fft_size
        self.three_mult = three_mult
        self.cc = [
            dict(select=0, predelay=0, speed=100, vsteps=0, dry=np.float32(0.5), wet=np.float32(0.5),
                 panDry=np.float32(0), panWet=np.float32(0), level=np.float32(1))
            for _ in range(2)
        ]
        self.ir = {}
        self.irfft = [[np.zeros(N, np.complex128) for _ in range(2)] for _ in range(2)]
        self.resid = [np.zeros(N + MAX_PREDELAY, np.complex128) for _ in range(2)]

    def prepare(self, idx, lr, nframes=1024):
        lr = np.asarray(lr, dtype=np.float32).reshape(-1, 2)
        n = min(len(lr), self.N - nframes)
        z = np.zeros(self.N, np.complex128)
        z[:n] = lr[:n, 0].astype(np.float64) + 1j * lr[:n, 1].astype(np.float64)
        self.ir[idx] = unpack(np.fft.fft(z))

    def _interp(self, half):
        cc = self.cc[half]
        N = self.N
        H = self.ir[cc["select"]]
        div = float(cc["vsteps"] + 5)
        for c in range(2):
            a = self.irfft[half][c]
            s = np.arange(0, N // 2)
            vv = a[s] + (H[c][s] * float(cc["wet"]) - a[s]) / div
            a[s] = vv
            a[N - s[1:]] = np.conj(vv[1:])
        if cc["vsteps"] > 0:
            cc["vsteps"] -= 1

    def _mul(self, a, b):
        if not self.three_mult:
            return a * b
        re = a.real * b.real - a.imag * b.imag
        im = (a.real + a.imag) * (b.real + b.imag) - re
        return re + 1j * im

    def process_block(self, in1, in2):
        N = self.N
        n = len(in1)
        c0, c1 = self.cc
        z = np.zeros(N, np.complex128)
        z[:n] = np.asarray(in1, np.float64) + 1j * np.asarray(in2, np.float64)
        self._interp(0)
        self._interp(1)
        X1, X2 = unpack(np.fft.fft(z))
        gl = [pan_l(float(c0["panWet"])) * float(c0["level"]) / N, pan_l(float(c1["panWet"])) * float(c1["level"]) / N]
        gr = [pan_r(float(c0["panWet"])) * float(c0["level"]) / N, pan_r(float(c1["panWet"])) * float(c1["level"]) / N]
        out = []
        pd = int(c0["predelay"])
        for c, g in ((0, gl), (1, gr)):
            Y = self._mul(X1, self.irfft[0][c]) * g[0] + self._mul(X2, self.irfft[1][c]) * g[1]
            w = np.fft.ifft(Y) * N  # unnormalised inverse
            acc = self.resid[c].copy()
            seg = acc[:N].copy()
            seg[pd:] += w[: N - pd]
            seg = np.clip(seg.real, -1, 1) + 1j * np.clip(seg.imag, -1, 1)
            acc[:N] = seg
            out.append(acc)
        dl = [float(c0["dry"]) * pan_l(float(c0["panDry"])) * float(c0["level"]),
              float(c1["dry"]) * pan_l(float(c1["panDry"])) * float(c1["level"])]
        dr = [float(c0["dry"]) * pan_r(float(c0["panDry"])) * float(c0["level"]),
              float(c1["dry"]) * pan_r(float(c1["panDry"])) * float(c1["level"])]
        x1, x2 = z[:n].real, z[:n].imag
        aL = x1 * dl[0] + x2 * dl[1]
        aR = x1 * dr[0] + x2 * dr[1]
        out[0][:n] += aL + 1j * aL
        out[1][:n] += aR + 1j * aR
        res = np.stack([out[0][:n].real, out[1][:n].real])
        for c in range(2):
            self.resid[c][: N + MAX_PREDELAY - n] = out[c][n:]
        return res

    def process(self, in1, in2, block=256):
        outs = [self.process_block(in1[b : b + block], in2[b : b + block]) for b in range(0, len(in1), block)]
        return np.concatenate(outs, axis=1)
