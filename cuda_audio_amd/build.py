"""Build libmcconv.so (hand-written HIP for gfx950) in-tree with hipcc.

The shared object is git-ignored but travels to the GPU box with the repo
snapshot; hipcc cross-compiles gfx950 without a GPU present.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "mcconv.hip")
DEPS = [SRC] + [os.path.join(HERE, "csrc", h) for h in ("kernels.hip.h", "jack_tail.hip.h", "fft512.hip.h", "ossave.hip.h", "singlefft.hip.h", "singlefft_host.hip.h", "params_handoff.h")] + [
    os.path.join(os.path.dirname(HERE), "include", "mcconv.h")]
OUT = os.path.join(HERE, "libmcconv.so")
FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-shared", "-fPIC", "-Wall", "-Wno-unused-function"]


def hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (set HIPCC)")


# the multi-GPU driver (include/mcconv_group.h): links the engine library and RCCL
GROUP_SRC = os.path.join(HERE, "csrc", "mcgroup.hip")
GROUP_OUT = os.path.join(HERE, "libmcconv_rccl.so")
GROUP_DEPS = [GROUP_SRC, os.path.join(os.path.dirname(HERE), "include", "mcconv_group.h"), os.path.join(os.path.dirname(HERE), "include", "mcconv.h")]


def build_group(force=False, verbose=False):
    if not force and os.path.exists(GROUP_OUT) and all(os.path.getmtime(d) <= os.path.getmtime(GROUP_OUT) for d in GROUP_DEPS + [OUT]):
        return GROUP_OUT
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    cmd = [hipcc()] + FLAGS + ["-o", GROUP_OUT, GROUP_SRC, "-L" + HERE, "-lmcconv", "-L" + os.path.join(rocm, "lib"), "-lrccl",
                               "-Wl,-rpath,$ORIGIN", "-Wl,-rpath," + os.path.join(rocm, "lib")]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return GROUP_OUT


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(d) > t for d in DEPS)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return OUT
    cmd = [hipcc()] + FLAGS + ["-o", OUT, SRC]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
    build_group(force="--force" in sys.argv, verbose=True)
    print(OUT)
    print(GROUP_OUT)
