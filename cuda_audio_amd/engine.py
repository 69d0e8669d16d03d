"""Python host mirror of the reference's `Convolution` class (src/conv.h:30-86)
over the C ABI of include/mcconv.h.

Same names and argument meaning as the reference for the hot-path surface:
`Convolution(name, fftSize)`, `cc[i].value.*`, `prepare(idx, wav, nframes=1024)`,
`onProcess`, `onMidiMessage`, `avgRuntime()`.  JACK port plumbing is replaced
by explicit buffers: `onProcess(in1, in2)` takes the two capture buffers and
returns the two playback buffers.  All arithmetic runs in libmcconv.so (HIP);
there is no CPU path here.
"""
import ctypes as C
import weakref

import numpy as np

from . import _lib
from ._lib import MC_BLOCK, McCcValue, McConfig, McKernelStats, check

CONV_DEFAULT_FFTSIZE = 512 * 256  # conv.h:10-12
CONV_MAX_SPEED = 1024             # conv.h:22-24
CONV_MAX_PREDELAY = 8192          # conv.h:26-28


def _f32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


class _CCValueView:
    """cc[i].value — attribute access backed by mc_get_params/mc_set_params."""

    _names = ("select", "predelay", "speed", "vsteps", "dry", "wet", "panDry", "panWet", "level")

    def __init__(self, eng, half):
        object.__setattr__(self, "_eng", eng)
        object.__setattr__(self, "_half", half)

    def _get(self):
        v = McCcValue()
        check(self._eng._L.mc_get_params(self._eng._h, self._half, C.byref(v)))
        return v

    def __getattr__(self, k):
        if k not in self._names:
            raise AttributeError(k)
        return getattr(self._get(), k)

    def __setattr__(self, k, val):
        if k not in self._names:
            raise AttributeError(k)
        v = self._get()
        setattr(v, k, val)
        check(self._eng._L.mc_set_params(self._eng._h, self._half, C.byref(v)))

    def update(self, **kw):
        v = self._get()
        for k, val in kw.items():
            if k not in self._names:
                raise AttributeError(k)
            setattr(v, k, val)
        check(self._eng._L.mc_set_params(self._eng._h, self._half, C.byref(v)))


class CC:
    """Convolution::CC (conv.h:33-50): controller numbers + current values."""

    def __init__(self, eng, half):
        self.device = None
        self.message = 176
        self.select = self.predelay = self.dry = self.wet = self.speed = 0
        self.panDry = self.panWet = self.level = 0
        self.value = _CCValueView(eng, half)

    def ccmap(self):
        return (self.select, self.predelay, self.dry, self.wet, self.speed, self.panDry, self.panWet, self.level)


class Convolution:
    def __init__(self, name="Conv", fftSize=CONV_DEFAULT_FFTSIZE, *, max_batch=256, device=-1, compat=True,
                 part_begin=0, part_end=0, max_partitions=0, stream_threshold=0, precision="fp32", period=256, pipeline=False,
                 form="partitioned"):
        self.name = name
        self._L = _lib.load()
        cfg = McConfig()
        self._L.mc_default_config(C.byref(cfg))
        cfg.device = device
        cfg.n_ref = fftSize
        cfg.max_batch = max_batch
        cfg.compat = 1 if compat else 0
        cfg.part_begin, cfg.part_end = part_begin, part_end
        cfg.max_partitions = max_partitions
        cfg.stream_threshold = stream_threshold
        cfg.precision = {"fp32": 0, "fp16": 1}[precision]
        cfg.period = period
        cfg.pipeline = 1 if pipeline else 0
        cfg.form = {"partitioned": 0, "single": 1}[form]
        h = C.c_void_p()
        check(self._L.mc_create(C.byref(cfg), C.byref(h)))
        self._h = h
        self.fftSize = fftSize
        self.max_batch = max_batch
        self.cc = [CC(self, 0), CC(self, 1)]

    # -- lifetime -----------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            self._L.mc_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self):
        check(self._L.mc_reset(self._h))

    def set_period(self, nframes):
        """JACK period the host calls onProcess with: 256, 512 or 1024 frames (resets the signal state)."""
        check(self._L.mc_set_period(self._h, nframes))

    # -- reference surface ----------------------------------------------------
    def prepare(self, idx, wav, nframes=1024):
        """Convolution::prepare (conv.cu:207-253).  `wav` is float32 [frames, 2]
        (what WavFile.buffer holds) or an object with a `.buffer` of that shape."""
        lr = _f32(getattr(wav, "buffer", wav)).reshape(-1, 2)
        check(self._L.mc_load_ir(self._h, idx, _fp(lr), lr.shape[0], nframes))

    def onProcess(self, in1, in2):
        """One JACK period (conv.cu:287-466): returns (L, R) float32 arrays."""
        in1, in2 = _f32(in1), _f32(in2)
        n = in1.shape[0]
        outL, outR = np.empty(n, np.float32), np.empty(n, np.float32)
        check(self._L.mc_process(self._h, _fp(in1), _fp(in2), _fp(outL), _fp(outR), n))
        return outL, outR

    def onMidiMessage(self, sender, buffer):
        """conv.cu:278-285: a 3-byte controller message from `sender`."""
        if len(buffer) < 3:
            return
        for half, cc in enumerate(self.cc):
            if cc.device is sender and cc.message == buffer[0]:
                arr = (C.c_uint8 * 8)(*cc.ccmap())
                check(self._L.mc_handle_cc(self._h, half, arr, buffer[1], buffer[2]))

    def avgRuntime(self):
        return self._L.mc_avg_runtime_ms(self._h)

    # -- throughput surface -----------------------------------------------------
    def process(self, in1, in2, out=None):
        """Consecutive blocks from host arrays (length a multiple of 256) in ONE mc_process_batch call; returns
        float32 [2, n].  The engine cuts long runs into chunks itself (16384 blocks through its pinned staging
        buffer for pageable arrays; its preferred batch, three streams, for pinned ones - `pinned_array`)."""
        in1, in2 = _f32(in1), _f32(in2)
        n = in1.shape[0]
        if n % MC_BLOCK or in2.shape[0] != n:
            raise ValueError("inputs must have equal length, a multiple of 256")
        if out is None:
            out = np.empty((2, n), np.float32)
        elif (not isinstance(out, np.ndarray) or out.dtype != np.float32 or out.shape != (2, n)
              or not out[0].flags.c_contiguous or not out[1].flags.c_contiguous or not out.flags.writeable):
            # the library writes n float32 values through each row pointer: anything else would be written past or across
            raise ValueError("out must be a writeable float32 array of shape (2, n) with C-contiguous rows")
        check(self._L.mc_process_batch(self._h, _fp(in1), _fp(in2), _fp(out[0]), _fp(out[1]), n // MC_BLOCK))
        return out

    def pinned_array(self, shape):
        """float32 array in pinned host memory (mc_host_alloc): buffers of this kind let mc_process_batch overlap
        copy-in, kernels and copy-out.  Freed with the array (keeps a reference to its allocation)."""
        n = int(np.prod(shape))
        p = self._L.mc_host_alloc(n * 4)
        if not p:
            raise MemoryError("mc_host_alloc failed")
        buf = (C.c_float * n).from_address(p)
        weakref.finalize(buf, self._L.mc_host_free, p)  # every view of the array keeps `buf` alive
        return np.frombuffer(buf, dtype=np.float32).reshape(shape)

    def process_device(self, d_in1, d_in2, d_outL, d_outR, nblocks):
        """Device pointers (ints, e.g. torch.Tensor.data_ptr()); asynchronous."""
        check(self._L.mc_process_batch_device(self._h, d_in1, d_in2, d_outL, d_outR, nblocks))

    def process_slice_device(self, d_in1, d_in2, d_outL, d_outR, nblocks, first, count):
        """Block-sliced operation: the same batch on every GPU, this engine finishes output blocks
        [first, first + count) into d_outL / d_outR (count * 256 floats each); asynchronous."""
        check(self._L.mc_process_batch_slice_device(self._h, d_in1, d_in2, d_outL, d_outR, nblocks, first, count))

    def partial_device(self, d_in1, d_in2, d_partial, nblocks):
        check(self._L.mc_partial_batch_device(self._h, d_in1, d_in2, d_partial, nblocks))

    def finish_device(self, d_in1, d_in2, d_wet_sum, d_outL, d_outR, nblocks):
        check(self._L.mc_finish_batch_device(self._h, d_in1, d_in2, d_wet_sum, d_outL, d_outR, nblocks))

    def finish_slice_device(self, d_in1, d_in2, d_wet_sum_slice, d_outL, d_outR, nblocks, first, count):
        """The finish after a reduce-scatter: d_wet_sum_slice = [L | R] of blocks [first, first + count) only."""
        check(self._L.mc_finish_batch_slice_device(self._h, d_in1, d_in2, d_wet_sum_slice, d_outL, d_outR, nblocks, first, count))

    def sync(self):
        check(self._L.mc_sync(self._h))

    def fence(self):
        """Pipelined engines: the engine's stream waits for every batch issued so far (outputs complete in stream
        order after this)."""
        check(self._L.mc_fence(self._h))

    def fence_older(self):
        """... every batch except the most recently issued one."""
        check(self._L.mc_fence_older(self._h))

    def set_stream(self, stream_ptr):
        """hipStream_t as an int; None / 0 = the engine's own non-blocking stream (NOT ordered with the default
        stream: use `use_torch_stream` when the buffers are torch tensors)."""
        check(self._L.mc_set_stream(self._h, stream_ptr))

    def use_torch_stream(self, stream=None):
        """Launch on `stream` (default: torch's current stream) so that the engine is ordered with the torch ops
        and collectives that produce and consume its device buffers.  torch reports the default stream as handle 0,
        which mc_set_stream reads as "own stream": MC_STREAM_DEFAULT (hipStreamLegacy) names it explicitly."""
        import torch

        ptr = (stream or torch.cuda.current_stream()).cuda_stream
        check(self._L.mc_set_stream(self._h, ptr if ptr else 1))

    # -- introspection ----------------------------------------------------------
    def num_irs(self):
        return self._L.mc_num_irs(self._h)

    def ir_info(self, idx):
        out = (C.c_double * 6)()
        check(self._L.mc_ir_info(self._h, idx, out))
        return dict(sigma=(out[0], out[1]), alpha=(out[2], out[3]), taps=int(out[4]), partitions=int(out[5]))

    def enable_kernel_timing(self, on=True):
        check(self._L.mc_enable_kernel_timing(self._h, 1 if on else 0))

    def kernel_stats(self, reset=False):
        ks = McKernelStats()
        check(self._L.mc_get_kernel_stats(self._h, C.byref(ks), 1 if reset else 0))
        return dict(launches=ks.launches, blocks=ks.blocks, total_ms=ks.total_ms, last_ms=ks.last_ms,
                    resident=bool(ks.resident), partitions=ks.partitions, fast_levels=ks.fast_levels)

    def algorithmic_bytes_per_block(self):
        return self._L.mc_algorithmic_bytes_per_block(self._h)

    def preferred_batch(self, at_most):
        """Batch length (blocks) <= at_most that wastes nothing of the second-level transform's chunks for the
        loaded IRs (mc_preferred_batch)."""
        return int(self._L.mc_preferred_batch(self._h, int(at_most)))

    def blocks_processed(self):
        return self._L.mc_blocks_processed(self._h)

    def park_stats(self):
        """JACK path: how many parked periods were used, gave up on their own (host away longer than the park time)
        and were told to give up (mc_debug_read item 6; host-side counters, no stream access)."""
        a = np.zeros(3, np.uint64)
        check(self._L.mc_debug_read(self._h, 6, 0, a.ctypes.data_as(C.c_void_p), 0, a.nbytes, None))
        return dict(used=int(a[0]), timed_out=int(a[1]), cancelled=int(a[2]))

    def tail_forms(self):
        """JACK path, 256-frame periods: tails by the form partition 0 took - frequency domain (the period was already there when
        the tail looked: calls back to back, periods launched on arrival) or time domain (a parked tail that had to wait).
        Counted by the kernel (mc_debug_read item 16; leaves the JACK path and waits for the stream)."""
        a = np.zeros(2, np.uint32)
        check(self._L.mc_debug_read(self._h, 16, 0, a.ctypes.data_as(C.c_void_p), 0, a.nbytes, None))
        return dict(frequency_domain=int(a[0]), time_domain=int(a[1]))

    def drop_stats(self):
        """Q8 regime: batches by the form their cut terms took (mc_debug_read item 9; host-side counters)."""
        a = np.zeros(4, np.uint64)
        check(self._L.mc_debug_read(self._h, 9, 0, a.ctypes.data_as(C.c_void_p), 0, a.nbytes, None))
        return dict(drop_fft=int(a[0]), forward_transforms=int(a[1]), tiles=int(a[2]), carried_periods=int(a[3]))

    def mac_stats(self):
        """Batch launches by the form their partition sums took (mc_debug_read item 10; host-side counters)."""
        a = np.zeros(3, np.uint64)
        check(self._L.mc_debug_read(self._h, 10, 0, a.ctypes.data_as(C.c_void_p), 0, a.nbytes, None))
        return dict(fused=int(a[0]), split=int(a[1]), resident=int(a[2]))

    def os_stats(self):
        """Batches that took the overlap-save form of long settled batches and builds of its spectra (mc_debug_read
        item 11; host-side counters)."""
        a = np.zeros(2, np.uint64)
        check(self._L.mc_debug_read(self._h, 11, 0, a.ctypes.data_as(C.c_void_p), 0, a.nbytes, None))
        return dict(batches=int(a[0]), spectra_builds=int(a[1]))

    def lab_build(self):
        """True when the loaded library was built with -DMCCONV_LAB: the measurement switches of rounds 1-3 and the alternative
        kernels exist only there (mc_debug_read item 15)."""
        a = np.zeros(1, np.uint64)
        check(self._L.mc_debug_read(self._h, 15, 0, a.ctypes.data_as(C.c_void_p), 0, a.nbytes, None))
        return bool(a[0])

    def param_generation(self, published=False):
        """Generation number of the parameter pair the last process call ran on (published=True: of the pair
        published last).  Every mc_set_params / mc_handle_cc publishes a new pair (csrc/params_handoff.h)."""
        a = np.zeros(1, np.uint64)
        check(self._L.mc_debug_read(self._h, 8 if published else 7, 0, a.ctypes.data_as(C.c_void_p), 0, a.nbytes, None))
        return int(a[0])

    def debug_dims(self):
        d = (C.c_uint64 * 4)()
        check(self._L.mc_debug_read(self._h, 1, 0, None, 0, 0, d))
        return dict(pstride=d[0], ring=d[1], max_batch=d[2], wet_ring=d[3])

    def debug_read(self, which, idx, dtype, offset_elems, count):
        a = np.empty(count, dtype=dtype)
        check(self._L.mc_debug_read(self._h, which, idx, a.ctypes.data_as(C.c_void_p), offset_elems * a.itemsize,
                                    a.nbytes, None))
        return a

    def ir_spectra(self, idx):
        """IR spectra as complex [2 ch][partitions][256 packed bins] (diagnostics)."""
        info, dims = self.ir_info(idx), self.debug_dims()
        ps = dims["pstride"]
        raw = self.debug_read(0, idx, np.float32, 0, 256 * ps * 4).reshape(256, ps, 4)
        P = info["partitions"]
        HL = raw[:, :P, 0] + 1j * raw[:, :P, 1]
        HR = raw[:, :P, 2] + 1j * raw[:, :P, 3]
        return np.stack([HL.T, HR.T])
