"""Build the plain-C part of the oracle (test infrastructure) with gcc: oracle/_build/libvq_oracle.so."""
import ctypes
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_build", "libvq_oracle.so")


def build(force=False):
    src = os.path.join(HERE, "vq_oracle.c")
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    if force or not os.path.exists(OUT) or os.path.getmtime(OUT) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-ffp-contract=off", "-o", OUT, src])
    return OUT


def load():
    lib = ctypes.CDLL(build())
    lib.vq_argmin_bf16_oracle.restype = None
    lib.vq_argmin_bf16_oracle.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                          ctypes.c_void_p, ctypes.c_void_p]
    return lib


def vq_argmin(z, codebook):
    """z, codebook: torch tensors (any float dtype; rounded to bf16 first). Returns (ids int64, gap fp32)."""
    import torch
    lib = load()
    zf = z.reshape(-1, z.shape[-1]).to(torch.bfloat16).float().contiguous()
    ef = codebook.to(torch.bfloat16).float().contiguous()
    ids = torch.empty(zf.shape[0], dtype=torch.int64)
    gap = torch.empty(zf.shape[0], dtype=torch.float32)
    lib.vq_argmin_bf16_oracle(zf.data_ptr(), ef.data_ptr(), zf.shape[0], ef.shape[0], ef.shape[1], ids.data_ptr(),
                              gap.data_ptr())
    return ids, gap


if __name__ == "__main__":
    print(build(force=True))
