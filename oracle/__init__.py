"""ctypes binding of the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg, never by the product package (cuda_audio_amd).
PARITY UNPINNED: see oracle/oracle.h.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")
_lib = None

BLOCK = 256


class CCValue(C.Structure):
    """Mirror of Convolution::CC::value (reference src/conv.h:38-49)."""

    _fields_ = [
        ("select", C.c_uint64),
        ("predelay", C.c_uint64),
        ("speed", C.c_uint64),
        ("vsteps", C.c_uint64),
        ("dry", C.c_float),
        ("wet", C.c_float),
        ("panDry", C.c_float),
        ("panWet", C.c_float),
        ("level", C.c_float),
    ]


def build():
    """Compile liboracle.so with the committed Makefile (gcc only)."""
    subprocess.check_call(["make", "-C", _HERE, "-s"])


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        build()
    L = C.CDLL(_LIB_PATH)
    fp = C.POINTER(C.c_float)
    dp = C.POINTER(C.c_double)
    L.orc_fft.argtypes = [dp, dp, C.c_size_t, C.c_int]
    L.orc_direct_conv.argtypes = [fp, C.c_size_t, fp, C.c_size_t, dp]
    L.orc_wav_decode_s16.argtypes = [C.POINTER(C.c_int16), C.c_size_t, fp]
    L.orc_wav_decode_s24.argtypes = [C.POINTER(C.c_uint8), C.c_size_t, fp]
    L.orc_handle_cc.argtypes = [C.POINTER(CCValue), C.POINTER(C.c_uint8), C.c_uint8, C.c_int, C.c_size_t]
    L.orc_ref_create.restype = C.c_void_p
    L.orc_ref_create.argtypes = [C.c_size_t, C.c_int]
    L.orc_ref_destroy.argtypes = [C.c_void_p]
    L.orc_ref_prepare.argtypes = [C.c_void_p, C.c_size_t, fp, C.c_size_t, C.c_size_t]
    L.orc_ref_cc.restype = C.POINTER(CCValue)
    L.orc_ref_cc.argtypes = [C.c_void_p, C.c_int]
    L.orc_ref_num_irs.restype = C.c_size_t
    L.orc_ref_num_irs.argtypes = [C.c_void_p]
    L.orc_ref_process.argtypes = [C.c_void_p, fp, fp, dp, dp, C.c_size_t]
    L.orc_ref_ir_sums.argtypes = [C.c_void_p, C.c_size_t, dp]
    L.orc_upols_create.restype = C.c_void_p
    L.orc_upols_create.argtypes = [C.c_size_t, C.c_int]
    L.orc_upols_destroy.argtypes = [C.c_void_p]
    L.orc_upols_prepare.argtypes = [C.c_void_p, C.c_size_t, fp, C.c_size_t, C.c_size_t]
    L.orc_upols_cc.restype = C.POINTER(CCValue)
    L.orc_upols_cc.argtypes = [C.c_void_p, C.c_int]
    L.orc_upols_process.argtypes = [C.c_void_p, fp, fp, dp, dp, C.c_size_t]
    L.orc_upols_set_shard.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t]
    L.orc_upols_partial.argtypes = [C.c_void_p, fp, fp, dp, dp]
    L.orc_upols_finish.argtypes = [C.c_void_p, fp, fp, dp, dp, dp, dp]
    L.orc_upols_range.argtypes = [C.c_void_p, fp, fp, C.c_size_t, C.c_size_t, dp, dp]
    L.orc_upols_range_settled.argtypes = [C.c_void_p, fp, fp, C.c_size_t, C.c_size_t, dp, dp]
    L.orc_cpu32_create.restype = C.c_void_p
    L.orc_cpu32_create.argtypes = [fp, fp, C.c_size_t]
    L.orc_cpu32_destroy.argtypes = [C.c_void_p]
    L.orc_cpu32_partitions.restype = C.c_size_t
    L.orc_cpu32_partitions.argtypes = [C.c_void_p]
    L.orc_cpu32_process.argtypes = [C.c_void_p, fp, fp, fp, fp, C.c_size_t, fp, fp, C.c_int]
    L.orc_max_threads.restype = C.c_int
    _lib = L
    return L


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def fft(x, sign=-1):
    """Unnormalised DFT of a complex vector with the oracle's own radix-2 FFT."""
    x = np.asarray(x, dtype=np.complex128)
    re = np.ascontiguousarray(x.real)
    im = np.ascontiguousarray(x.imag)
    lib().orc_fft(_dp(re), _dp(im), len(x), sign)
    return re + 1j * im


def direct_conv(x, h):
    x, h = _f32(x), _f32(h)
    y = np.zeros(len(x) + len(h) - 1, dtype=np.float64)
    lib().orc_direct_conv(_fp(x), len(x), _fp(h), len(h), _dp(y))
    return y


def wav_decode_s16(lr_int16):
    a = np.ascontiguousarray(lr_int16, dtype=np.int16).reshape(-1)
    out = np.zeros(a.size, dtype=np.float32)
    lib().orc_wav_decode_s16(a.ctypes.data_as(C.POINTER(C.c_int16)), a.size // 2, _fp(out))
    return out.reshape(-1, 2)


def wav_decode_s24(raw_bytes):
    a = np.frombuffer(bytes(raw_bytes), dtype=np.uint8).copy()
    frames = a.size // 6
    out = np.zeros(frames * 2, dtype=np.float32)
    lib().orc_wav_decode_s24(a.ctypes.data_as(C.POINTER(C.c_uint8)), frames, _fp(out))
    return out.reshape(-1, 2)


class _Engine:
    """Shared driver for the refcompat and upols state machines."""

    _create = _destroy = _prepare = _cc = _process = None

    def __init__(self, *args):
        self._h = getattr(lib(), self._create)(*args)
        if not self._h:
            raise MemoryError("oracle engine allocation failed")

    def close(self):
        if self._h:
            getattr(lib(), self._destroy)(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def prepare(self, idx, lr, nframes=1024):
        """Convolution::prepare (conv.cu:207-253); lr is float32 [frames, 2]."""
        lr = _f32(lr).reshape(-1, 2)
        rc = getattr(lib(), self._prepare)(self._h, idx, _fp(lr), lr.shape[0], nframes)
        if rc:
            raise ValueError("oracle prepare failed")

    def cc(self, half):
        return getattr(lib(), self._cc)(self._h, half).contents

    def set(self, half=None, **kw):
        halves = (0, 1) if half is None else (half,)
        for h in halves:
            v = self.cc(h)
            for k, val in kw.items():
                setattr(v, k, val)

    def process(self, in1, in2, block=BLOCK):
        """Run consecutive `block`-frame periods (256 unless stated); returns float64 [2, n]."""
        in1, in2 = _f32(in1), _f32(in2)
        n = len(in1)
        assert n % block == 0 and len(in2) == n
        out = np.zeros((2, n), dtype=np.float64)
        f = getattr(lib(), self._process)
        for b in range(n // block):
            s = slice(b * block, (b + 1) * block)
            a, bb = in1[s], in2[s]
            f(self._h, _fp(a), _fp(bb), _dp(out[0, s]), _dp(out[1, s]), block)
        return out


class RefCompat(_Engine):
    """float64 restatement of the reference's single-FFT algorithm."""

    _create, _destroy = "orc_ref_create", "orc_ref_destroy"
    _prepare, _cc, _process = "orc_ref_prepare", "orc_ref_cc", "orc_ref_process"

    def __init__(self, fft_size, three_mult=True):
        super().__init__(fft_size, 1 if three_mult else 0)
        self.fft_size = fft_size

    def ir_sums(self, idx):
        out = np.zeros(4)
        if lib().orc_ref_ir_sums(self._h, idx, _dp(out)):
            raise KeyError(idx)
        return out

    def num_irs(self):
        return lib().orc_ref_num_irs(self._h)


class Upols(_Engine):
    """float64 uniform-partition overlap-save form."""

    _create, _destroy = "orc_upols_create", "orc_upols_destroy"
    _prepare, _cc, _process = "orc_upols_prepare", "orc_upols_cc", "orc_upols_process"

    def __init__(self, n_ref, compat=True, part_begin=0, part_end=0):
        super().__init__(n_ref, 1 if compat else 0)
        lib().orc_upols_set_shard(self._h, part_begin, part_end)

    def partial(self, in1, in2):
        """This shard's wet signal (pre-predelay) for consecutive blocks: float64 [2, n]."""
        in1, in2 = _f32(in1), _f32(in2)
        n = len(in1)
        assert n == BLOCK, "partial/finish work one block at a time"
        w = np.zeros((2, n), dtype=np.float64)
        lib().orc_upols_partial(self._h, _fp(in1), _fp(in2), _dp(w[0]), _dp(w[1]))
        return w

    def finish(self, in1, in2, wsum):
        in1, in2 = _f32(in1), _f32(in2)
        wsum = np.ascontiguousarray(wsum, dtype=np.float64)
        out = np.zeros((2, BLOCK), dtype=np.float64)
        lib().orc_upols_finish(self._h, _fp(in1), _fp(in2), _dp(wsum[0]), _dp(wsum[1]), _dp(out[0]), _dp(out[1]))
        return out


    def range(self, in1, in2, b0, n, settled=False):
        """Output blocks [b0, b0 + n) of a stream that started cold at block 0 under the parameters now set
        (constant throughout); in1 / in2 hold at least blocks [0, b0 + n).  float64 [2, n * 256].  The blocks
        before b0 contribute gains, Q1/Q2 terms and spectra; their partition sums are not run.
        settled=True: the buffers are an excerpt of a long stream in steady state (cross-fade converged) and b0
        counts from their start; it must lie beyond one reference length / the longest IR + the predelay."""
        in1, in2 = _f32(in1), _f32(in2)
        assert len(in1) >= (b0 + n) * BLOCK and len(in2) >= (b0 + n) * BLOCK
        out = np.zeros((2, n * BLOCK), dtype=np.float64)
        f = lib().orc_upols_range_settled if settled else lib().orc_upols_range
        rc = f(self._h, _fp(in1), _fp(in2), b0, n, _dp(out[0]), _dp(out[1]))
        if rc:
            raise ValueError("orc_upols_range failed (%d)" % rc)
        return out


class Cpu32:
    """float32 OpenMP partitioned convolver — the timed CPU baseline."""

    def __init__(self, lr0, lr1):
        lr0, lr1 = _f32(lr0).reshape(-1, 2), _f32(lr1).reshape(-1, 2)
        assert lr0.shape == lr1.shape
        self._h = lib().orc_cpu32_create(_fp(lr0), _fp(lr1), lr0.shape[0])
        self.partitions = lib().orc_cpu32_partitions(self._h)

    def process(self, in1, in2, wet_gain, dry_gain, nthreads=0):
        in1, in2 = _f32(in1), _f32(in2)
        n = len(in1)
        assert n % BLOCK == 0
        out = np.zeros((2, n), dtype=np.float32)
        g, d = _f32(wet_gain).reshape(4), _f32(dry_gain).reshape(4)
        lib().orc_cpu32_process(self._h, _fp(in1), _fp(in2), _fp(out[0]), _fp(out[1]), n // BLOCK, _fp(g), _fp(d), nthreads)
        return out

    def close(self):
        if self._h:
            lib().orc_cpu32_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def max_threads():
    return lib().orc_max_threads()


def handle_cc(value, ccmap, m2, val, nb):
    """handleCC (conv.cu:255-276) on a CCValue; ccmap = 8 controller numbers."""
    arr = (C.c_uint8 * 8)(*ccmap)
    lib().orc_handle_cc(C.byref(value), arr, m2, val, nb)
