"""Build libeoc_tfhe_gpu.so in-tree (hipcc cross-compiles gfx950 without a GPU).

  python -m eoc_tfhe_amd.build [--force] [--verbose]

-ffp-contract=off is part of the arithmetic contract of the canonical transform:
every fused multiply-add in the kernels is written explicitly.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libeoc_tfhe_gpu.so")
OBJ = os.path.join(HERE, "_build")

HIPCC = os.environ.get("HIPCC") or shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
GXX = os.environ.get("CXX") or shutil.which("g++") or "g++"
ARCH = "gfx950"

HIP_FLAGS = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", f"--offload-arch={ARCH}",
             "-Wall", "-Wno-unused-function", "-Wno-unused-value", "-Wno-unused-result"]
CXX_FLAGS = ["-O2", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fopenmp", "-mavx2", "-mfma", "-Wall"]


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _run(cmd, verbose):
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)


def build_stamps(verbose=False):
    """Diagnostic variant with in-kernel s_memtime stamps (tools/stamps.py); never the shipped library."""
    os.makedirs(OBJ, exist_ok=True)
    out = os.path.join(HERE, "libeoc_tfhe_gpu_stamps.so")
    eng_obj = os.path.join(OBJ, "engine_stamps.o")
    host_obj = os.path.join(OBJ, "host.o")
    leg_obj = os.path.join(OBJ, "legacy.o")
    extra = os.environ.get("EOC_EXTRA_HIP_FLAGS", "").split()
    _run([HIPCC, *HIP_FLAGS, "-DEOC_STAMPS", *extra, "-c", os.path.join(CSRC, "engine.hip"), "-o", eng_obj], verbose)
    if not os.path.exists(host_obj):
        _run([GXX, *CXX_FLAGS, "-c", os.path.join(CSRC, "host.cpp"), "-o", host_obj], verbose)
    if not os.path.exists(leg_obj):
        _run([GXX, *CXX_FLAGS, "-c", os.path.join(CSRC, "legacy.cpp"), "-o", leg_obj], verbose)
    mul_obj = os.path.join(OBJ, "multi.o")
    if not os.path.exists(mul_obj):
        _run([HIPCC, *HIP_FLAGS, "-c", os.path.join(CSRC, "multi.hip"), "-o", mul_obj], verbose)
    _run([HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", eng_obj, mul_obj, host_obj, leg_obj, "-o", out,
          "-lgomp", "-ldl", "-Wl,-rpath,/opt/rocm/lib"], verbose)
    return out


def build(force=False, verbose=False, extra_hip_flags=()):
    os.makedirs(OBJ, exist_ok=True)
    hdrs = [os.path.join(CSRC, f) for f in ("kernels.hip.h", "canon_twiddles.h", "common.h")]
    hdrs.append(os.path.join(ROOT, "include", "eoc_tfhe_gpu.h"))
    eng_src, eng_obj = os.path.join(CSRC, "engine.hip"), os.path.join(OBJ, "engine.o")
    mul_src, mul_obj = os.path.join(CSRC, "multi.hip"), os.path.join(OBJ, "multi.o")
    host_src, host_obj = os.path.join(CSRC, "host.cpp"), os.path.join(OBJ, "host.o")
    leg_src, leg_obj = os.path.join(CSRC, "legacy.cpp"), os.path.join(OBJ, "legacy.o")
    hdrs.append(os.path.join(CSRC, "host_internal.h"))
    if force or _newer(eng_obj, [eng_src] + hdrs):
        _run([HIPCC, *HIP_FLAGS, *extra_hip_flags, "-c", eng_src, "-o", eng_obj], verbose)
    if force or _newer(mul_obj, [mul_src] + hdrs):
        _run([HIPCC, *HIP_FLAGS, "-c", mul_src, "-o", mul_obj], verbose)
    if force or _newer(host_obj, [host_src] + hdrs):
        _run([GXX, *CXX_FLAGS, "-c", host_src, "-o", host_obj], verbose)
    if force or _newer(leg_obj, [leg_src] + hdrs):
        _run([GXX, *CXX_FLAGS, "-c", leg_src, "-o", leg_obj], verbose)
    if force or _newer(LIB, [eng_obj, mul_obj, host_obj, leg_obj]):
        _run([HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", eng_obj, mul_obj, host_obj, leg_obj, "-o", LIB,
              "-lgomp", "-ldl", "-Wl,-rpath,/opt/rocm/lib"], verbose)
    return LIB


if __name__ == "__main__":
    if "--stamps" in sys.argv:
        print(build_stamps(verbose=True))
        sys.exit(0)
    print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv or True))
