#!/usr/bin/env python
"""Run one GEMM shape/kernel a few times (for rocprofv3 --pmc passes):  gemm_one.py <variant> <M> <N> <K> [iters]
FOLD=1 (default): the launch seedmi_tokenize issues for the ViT QKV GEMM - LayerNorm folded in (seedmi_gemm_bf16_ext, BIAS epilogue,
kernel gemm256_kernel<1, true>); FOLD=0: the plain nn.Linear form (seedmi_gemm_bf16, gemm256_kernel<1, false>)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from seed_amd import lib as L  # noqa: E402

lib = L.load()
v, M, N, K = (int(a) for a in sys.argv[1:5])
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 3
g = torch.Generator(device="cuda").manual_seed(0)
A = torch.randn(M, K, device="cuda", generator=g).bfloat16()
W = (torch.randn(N, K, device="cuda", generator=g) * 0.02).bfloat16()
bias = torch.zeros(N, device="cuda").bfloat16()
C = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
L.check(lib.seedmi_set_option(b"gemm", v), "opt")
fold = os.environ.get("FOLD", "1") == "1" and N % 64 == 0
EPI = int(os.environ.get("EPI", str(L.EPI_BIAS)))       # 1 = BIAS, 2 = BIAS_GELU (the fc1 launch)
if fold:
    stats = torch.zeros(M + (M & 1), 2, dtype=torch.float32, device="cuda")
    L.check(lib.seedmi_layernorm_stats_bf16(L.ptr(A), K, M, K, 1e-6, L.ptr(stats), L.stream_ptr()), "stats")
    cs, b32 = W.float().sum(1).contiguous(), bias.float().contiguous()
    ext = L.GemmExt(L.ptr(stats), L.ptr(cs), L.ptr(b32), None, 0)
for _ in range(iters):
    if fold:
        L.check(lib.seedmi_gemm_bf16_ext(M, N, K, L.ptr(A), K, L.ptr(W), K, None, None, 0, EPI, L.ptr(C), N, 0, 0, ctypes.byref(ext),
                                         None, 0, L.stream_ptr()), "gemm ext")
    else:
        L.check(lib.seedmi_gemm_bf16(M, N, K, L.ptr(A), K, L.ptr(W), K, L.ptr(bias), None, 0, EPI, L.ptr(C), N, 0, 0,
                                     L.stream_ptr()), "gemm")
torch.cuda.synchronize()
