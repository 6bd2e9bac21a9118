#!/usr/bin/env python
"""Sweep the GEMM launch options (persistent on/off, L2 group height) on one shape: M N K from argv."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from seed_amd import lib as L  # noqa: E402

lib = L.load()
M, N, K = (int(x) for x in sys.argv[1:4])
A = torch.randn(M, K, device="cuda").bfloat16()
W = (torch.randn(N, K, device="cuda") * 0.02).bfloat16()
bias = torch.randn(N, device="cuda").bfloat16()
C = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
L.check(lib.seedmi_set_option(b"gemm", 256), "opt")
for persist in (1, 0):
    for gm in (1, 2, 4, 8, 16, 32):
        L.check(lib.seedmi_set_option(b"gemm_persist", persist), "opt")
        L.check(lib.seedmi_set_option(b"gemm_group_m", gm), "opt")
        ts = []
        for r in range(6):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            L.check(lib.seedmi_gemm_bf16(M, N, K, L.ptr(A), K, L.ptr(W), K, L.ptr(bias), None, 0, L.EPI_BIAS, L.ptr(C), N, 0, 0,
                                         L.stream_ptr()), "gemm")
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        med = sorted(ts[1:])[2]
        print("persist", persist, "group_m", gm, "ms %.4f" % med, "TF %.1f" % (2.0 * M * N * K / med / 1e9), flush=True)
