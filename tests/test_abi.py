"""C-ABI surface checks that need no GPU: the library loads and exports exactly
the entry points include/mcconv.h declares, with matching struct layouts."""
import ctypes as C
import os
import re

from cuda_audio_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header="mcconv.h"):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mc_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    L = _lib.load()
    declared = _declared()
    assert len(declared) >= 25
    missing = [s for s in declared if not hasattr(L, s)]
    assert not missing, missing
    assert sorted(_lib.SYMBOLS) == declared


def test_group_library_exports_every_declared_symbol():
    """libmcconv_rccl.so (the native multi-GPU driver, include/mcconv_group.h) loads - it links libmcconv.so and librccl.so -
    and exports what its header declares; argument errors come back without a GPU."""
    from cuda_audio_amd import group

    L = group.load()
    declared = _declared("mcconv_group.h")
    assert len(declared) == 10
    assert not [s for s in declared if not hasattr(L, s)]
    assert sorted(group.SYMBOLS) == declared
    h = C.c_void_p()
    assert L.mc_group_create(None, None, 0, C.byref(h)) == -1
    assert b"bad argument" in L.mc_group_last_error()
    assert L.mc_group_size(None) == 0 and L.mc_group_exchange(None) == b"none"


def test_struct_layouts_and_defaults():
    L = _lib.load()
    assert L.mc_abi_version() == 1
    assert C.sizeof(_lib.McConfig) == 64
    assert C.sizeof(_lib.McCcValue) == 56
    cfg = _lib.McConfig()
    L.mc_default_config(C.byref(cfg))
    assert cfg.struct_size == C.sizeof(_lib.McConfig)
    assert cfg.n_ref == 512 * 256 and cfg.compat == 1  # CONV_DEFAULT_FFTSIZE, conv.h:10-12
    v = _lib.McCcValue()
    L.mc_default_params(C.byref(v))
    # defaults of Convolution::CC::value, conv.h:38-49
    assert (v.select, v.predelay, v.speed, v.vsteps) == (0, 0, 100, 0)
    assert (v.dry, v.wet, v.panDry, v.panWet, v.level) == (0.5, 0.5, 0.0, 0.0, 1.0)


def test_argument_errors_do_not_need_a_gpu():
    L = _lib.load()
    assert L.mc_create(None, None) == -1
    assert b"null" in L.mc_last_error()
    cfg = _lib.McConfig()
    L.mc_default_config(C.byref(cfg))
    cfg.struct_size = 3
    h = C.c_void_p()
    assert L.mc_create(C.byref(cfg), C.byref(h)) == -1
    assert b"size mismatch" in L.mc_last_error()
    L.mc_default_config(C.byref(cfg))
    cfg.n_ref = 5000  # not a power of two
    assert L.mc_create(C.byref(cfg), C.byref(h)) == -1


def test_product_does_not_import_the_oracle():
    """The product package must not reference oracle/ (checker only)."""
    pkg = os.path.join(ROOT, "cuda_audio_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".hpp")):
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert "import oracle" not in text and "from oracle" not in text and "liboracle" not in text, f
