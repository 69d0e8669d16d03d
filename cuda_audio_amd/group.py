"""ctypes binding of libmcconv_rccl.so - the native multi-GPU driver declared in include/mcconv_group.h: IR partitions
sharded over the listed devices, one host thread per device, RCCL reduce-scatter of the partial wet blocks (a sum kernel
when one device is listed several times: virtual ranks on a one-GPU box).  No CPU path: a missing library raises."""
import ctypes as C
import os

import numpy as np

from . import _lib
from ._lib import MC_BLOCK, McCcValue, McConfig

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libmcconv_rccl.so")

SYMBOLS = ["mc_group_create", "mc_group_destroy", "mc_group_size", "mc_group_engine", "mc_group_shard", "mc_group_load_ir",
           "mc_group_set_params", "mc_group_process_batch", "mc_group_exchange", "mc_group_last_error"]

_L = None


class GroupError(RuntimeError):
    pass


def load():
    global _L
    if _L is not None:
        return _L
    _lib.load()  # the engine library first (the same copy the group library links)
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: run `python -m cuda_audio_amd.build` (there is no CPU path)")
    L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    fp = C.POINTER(C.c_float)
    L.mc_group_create.argtypes = [C.POINTER(McConfig), C.POINTER(C.c_int32), C.c_uint32, C.POINTER(C.c_void_p)]
    L.mc_group_create.restype = C.c_int
    L.mc_group_destroy.argtypes = [C.c_void_p]
    L.mc_group_destroy.restype = None
    L.mc_group_size.argtypes = [C.c_void_p]
    L.mc_group_size.restype = C.c_uint32
    L.mc_group_engine.argtypes = [C.c_void_p, C.c_uint32]
    L.mc_group_engine.restype = C.c_void_p
    L.mc_group_shard.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.mc_group_shard.restype = C.c_int
    L.mc_group_load_ir.argtypes = [C.c_void_p, C.c_uint64, fp, C.c_uint64, C.c_uint64]
    L.mc_group_load_ir.restype = C.c_int
    L.mc_group_set_params.argtypes = [C.c_void_p, C.c_int, C.POINTER(McCcValue)]
    L.mc_group_set_params.restype = C.c_int
    L.mc_group_process_batch.argtypes = [C.c_void_p, fp, fp, fp, fp, C.c_uint64]
    L.mc_group_process_batch.restype = C.c_int
    L.mc_group_exchange.argtypes = [C.c_void_p]
    L.mc_group_exchange.restype = C.c_char_p
    L.mc_group_last_error.argtypes = []
    L.mc_group_last_error.restype = C.c_char_p
    _L = L
    return L


def _check(rc):
    if rc != 0:
        raise GroupError(f"mc_group status {rc}: {load().mc_group_last_error().decode(errors='replace')}")


class ConvolutionGroup:
    """`Convolution` over several devices (no reference equivalent: gpu.cu:38-90 selects one device)."""

    def __init__(self, fftSize, devices, *, max_batch=256, compat=True, period=256, solo_exchange=False):
        self._L = load()
        E = _lib.load()
        cfg = McConfig()
        E.mc_default_config(C.byref(cfg))
        cfg.n_ref = fftSize
        cfg.max_batch = max_batch
        cfg.compat = 1 if compat else 0
        cfg.period = period
        cfg.reserved = 1 if solo_exchange else 0  # (one device: still through the RCCL exchange, see mcconv_group.h)
        devs = (C.c_int32 * len(devices))(*devices)
        h = C.c_void_p()
        _check(self._L.mc_group_create(C.byref(cfg), devs, len(devices), C.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._L.mc_group_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def size(self):
        return int(self._L.mc_group_size(self._h))

    def exchange(self):
        return self._L.mc_group_exchange(self._h).decode()

    def shard(self, rank):
        b, e = C.c_uint32(), C.c_uint32()
        _check(self._L.mc_group_shard(self._h, rank, C.byref(b), C.byref(e)))
        return int(b.value), int(e.value)

    def prepare(self, idx, wav, nframes=1024):
        lr = np.ascontiguousarray(getattr(wav, "buffer", wav), dtype=np.float32).reshape(-1, 2)
        _check(self._L.mc_group_load_ir(self._h, idx, lr.ctypes.data_as(C.POINTER(C.c_float)), lr.shape[0], nframes))

    def set_params(self, half, **kw):
        E = _lib.load()
        v = McCcValue()
        E.mc_default_params(C.byref(v))
        for k, val in kw.items():
            setattr(v, k, val)
        _check(self._L.mc_group_set_params(self._h, half, C.byref(v)))

    def process(self, in1, in2):
        in1 = np.ascontiguousarray(in1, dtype=np.float32)
        in2 = np.ascontiguousarray(in2, dtype=np.float32)
        n = in1.shape[0]
        if n % MC_BLOCK or in2.shape[0] != n:
            raise ValueError("inputs must have equal length, a multiple of 256")
        out = np.empty((2, n), np.float32)
        fp = C.POINTER(C.c_float)
        _check(self._L.mc_group_process_batch(self._h, in1.ctypes.data_as(fp), in2.ctypes.data_as(fp), out[0].ctypes.data_as(fp),
                                              out[1].ctypes.data_as(fp), n // MC_BLOCK))
        return out
