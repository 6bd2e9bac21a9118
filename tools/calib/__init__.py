"""libseedcal.so: calibration micro-benchmarks (MFMA-only loop, HBM stream read).  Measurement infrastructure for bench.py and the
tools/ scripts - not the product library, not the drop-in boundary."""
import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libseedcal.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
_lib = None


def build(force=False):
    src = os.path.join(HERE, "calib.hip")
    newest = max(os.path.getmtime(src), os.path.getmtime(os.path.join(HERE, "seedcal.h")))
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < newest:
        subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-o", LIB, src])
    return LIB


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            raise RuntimeError(f"{LIB} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'`")
        lib = C.CDLL(LIB)
        lib.seedcal_stream_read.restype = C.c_int
        lib.seedcal_stream_read.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]
        lib.seedcal_mfma_bf16.restype = C.c_int
        lib.seedcal_mfma_bf16.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_void_p]
        _lib = lib
    return _lib


def check(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed with code {rc}")
