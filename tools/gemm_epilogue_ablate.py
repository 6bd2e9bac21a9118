#!/usr/bin/env python
"""What the epilogue of the 256x256 GEMM costs (back-to-back launches): full kernel, no epilogue at all, epilogue without its
stores, and ordinary (L2-allocating) instead of streaming stores.  Timing only - masks 32/33 leave C unwritten."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from seed_amd import lib as L  # noqa: E402

lib = L.load()
NAMES = {0: "full", 32: "no epilogue", 33: "epilogue without stores", 34: "L2-allocating stores", 35: "streaming, no transpose"}
for name, M, N, K in [("qkv", 65792, 4224, 1408), ("fc1", 65792, 6144, 1408), ("fc2", 65792, 1408, 6144)]:
    A = torch.randn(M, K, device="cuda").bfloat16()
    W = (torch.randn(N, K, device="cuda") * 0.02).bfloat16()
    bias = torch.randn(N, device="cuda").bfloat16()
    C = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for mask in (0, 35, 34, 0, 35, 34, 33):
        L.check(lib.seedmi_set_option(b"gemm_ablate", mask), "opt")
        fn = lambda: L.check(lib.seedmi_gemm_bf16(M, N, K, L.ptr(A), K, L.ptr(W), K, L.ptr(bias), None, 0, L.EPI_BIAS, L.ptr(C), N,
                                                  0, 0, L.stream_ptr()), "gemm")
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(60):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 60
        print(name, NAMES[mask].ljust(26), "ms %.4f  TF %.1f" % (ms, 2.0 * M * N * K / ms / 1e9), flush=True)
lib.seedmi_set_option(b"gemm_ablate", 0)
