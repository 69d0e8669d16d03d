"""ctypes binding of libmcconv.so — the C ABI declared in include/mcconv.h.

The product path has no CPU fallback: if the HIP library is missing or fails
to load, importing an engine raises.  (The CPU oracle lives under oracle/ and
is never imported from here.)
"""
import ctypes as C
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
# MCCONV_LIB: another build of the same library (A/B measurements of kernel variants); there is still no CPU path
LIB_PATH = os.environ.get("MCCONV_LIB") or os.path.join(HERE, "libmcconv.so")

MC_BLOCK = 256
MC_MAX_PREDELAY = 8192

# every symbol include/mcconv.h declares (checked by tests/test_abi.py)
SYMBOLS = [
    "mc_abi_version", "mc_last_error", "mc_default_config", "mc_default_params", "mc_create", "mc_destroy",
    "mc_reset", "mc_set_period", "mc_load_ir", "mc_num_irs", "mc_ir_info", "mc_set_params", "mc_get_params", "mc_handle_cc",
    "mc_process", "mc_process_batch", "mc_process_batch_device", "mc_partial_batch_device",
    "mc_finish_batch_device", "mc_finish_batch_slice_device", "mc_process_batch_slice_device", "mc_sync", "mc_fence", "mc_fence_older", "mc_set_stream", "mc_get_stream", "mc_avg_runtime_ms",
    "mc_enable_kernel_timing", "mc_get_kernel_stats", "mc_algorithmic_bytes_per_block", "mc_blocks_processed", "mc_preferred_batch",
    "mc_debug_read", "mc_host_alloc", "mc_host_free",
]


class McConfig(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("device", C.c_int32),
        ("n_ref", C.c_uint64),
        ("max_batch", C.c_uint32),
        ("max_partitions", C.c_uint32),
        ("compat", C.c_uint32),
        ("part_begin", C.c_uint32),
        ("part_end", C.c_uint32),
        ("stream_threshold", C.c_uint32),
        ("precision", C.c_uint32),
        ("period", C.c_uint32),
        ("pipeline", C.c_uint32),
        ("form", C.c_uint32),
        ("reserved", C.c_uint32),
    ]


class McCcValue(C.Structure):
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


class McKernelStats(C.Structure):
    _fields_ = [
        ("launches", C.c_uint64),
        ("blocks", C.c_uint64),
        ("total_ms", C.c_double),
        ("last_ms", C.c_double),
        ("resident", C.c_uint32),
        ("partitions", C.c_uint32),
        ("fast_levels", C.c_uint32),
        ("reserved", C.c_uint32),
    ]


_lib = None


def load():
    """Load libmcconv.so; raise (never fall back) when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -m cuda_audio_amd.build` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback."
        )
    # One HIP runtime per process: PyTorch-ROCm wheels bundle their own
    # libamdhip64/libhsa-runtime64.  If libmcconv.so is loaded first it binds to
    # /opt/rocm's copies and a later `import torch` finds "No HIP GPUs"; loaded
    # after torch it binds (by soname) to torch's runtime, and stream handles
    # and device pointers interoperate.  So when torch is installed, load it
    # first.  Hosts without Python/torch (cuda_audio_amd/host) use /opt/rocm's.
    if "torch" not in sys.modules and os.environ.get("MCCONV_NO_TORCH_PRELOAD") != "1":
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    L = C.CDLL(LIB_PATH)
    vp, fp, u64 = C.c_void_p, C.POINTER(C.c_float), C.c_uint64
    L.mc_abi_version.restype = C.c_uint32
    L.mc_last_error.restype = C.c_char_p
    L.mc_default_config.argtypes = [C.POINTER(McConfig)]
    L.mc_default_params.argtypes = [C.POINTER(McCcValue)]
    L.mc_create.argtypes = [C.POINTER(McConfig), C.POINTER(vp)]
    L.mc_destroy.argtypes = [vp]
    L.mc_destroy.restype = None
    L.mc_reset.argtypes = [vp]
    L.mc_set_period.argtypes = [vp, C.c_uint32]
    L.mc_load_ir.argtypes = [vp, u64, fp, u64, u64]
    L.mc_num_irs.argtypes = [vp]
    L.mc_ir_info.argtypes = [vp, u64, C.POINTER(C.c_double)]
    L.mc_set_params.argtypes = [vp, C.c_int, C.POINTER(McCcValue)]
    L.mc_get_params.argtypes = [vp, C.c_int, C.POINTER(McCcValue)]
    L.mc_handle_cc.argtypes = [vp, C.c_int, C.POINTER(C.c_uint8), C.c_uint8, C.c_int]
    L.mc_process.argtypes = [vp, fp, fp, fp, fp, u64]
    L.mc_process_batch.argtypes = [vp, fp, fp, fp, fp, u64]
    L.mc_process_batch_device.argtypes = [vp, vp, vp, vp, vp, u64]
    L.mc_process_batch_slice_device.argtypes = [vp, vp, vp, vp, vp, u64, u64, u64]
    L.mc_partial_batch_device.argtypes = [vp, vp, vp, vp, u64]
    L.mc_finish_batch_device.argtypes = [vp, vp, vp, vp, vp, vp, u64]
    L.mc_finish_batch_slice_device.argtypes = [vp, vp, vp, vp, vp, vp, u64, u64, u64]
    L.mc_sync.argtypes = [vp]
    L.mc_fence.argtypes = [vp]
    L.mc_fence_older.argtypes = [vp]
    L.mc_set_stream.argtypes = [vp, vp]
    L.mc_get_stream.argtypes = [vp]
    L.mc_get_stream.restype = vp
    L.mc_avg_runtime_ms.argtypes = [vp]
    L.mc_avg_runtime_ms.restype = C.c_double
    L.mc_enable_kernel_timing.argtypes = [vp, C.c_int]
    L.mc_get_kernel_stats.argtypes = [vp, C.POINTER(McKernelStats), C.c_int]
    L.mc_algorithmic_bytes_per_block.argtypes = [vp]
    L.mc_algorithmic_bytes_per_block.restype = u64
    L.mc_blocks_processed.argtypes = [vp]
    L.mc_blocks_processed.restype = u64
    L.mc_preferred_batch.argtypes = [vp, u64]
    L.mc_preferred_batch.restype = u64
    L.mc_debug_read.argtypes = [vp, C.c_int, u64, vp, u64, u64, C.POINTER(u64)]
    L.mc_host_alloc.argtypes = [C.c_size_t]
    L.mc_host_alloc.restype = vp
    L.mc_host_free.argtypes = [vp]
    L.mc_host_free.restype = None
    _lib = L
    return L


class McError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"mcconv error {code}: {msg}")
        self.code = code


def check(rc):
    if rc != 0:
        raise McError(rc, load().mc_last_error().decode("utf-8", "replace"))
