#!/usr/bin/env python
"""Run one GEMM shape/kernel a few times (for rocprofv3 --pmc passes):  gemm_one.py <variant> <M> <N> <K> [iters]"""
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
for _ in range(iters):
    L.check(lib.seedmi_gemm_bf16(M, N, K, L.ptr(A), K, L.ptr(W), K, L.ptr(bias), None, 0, L.EPI_BIAS, L.ptr(C), N, 0, 0,
                                 L.stream_ptr()), "gemm")
torch.cuda.synchronize()
