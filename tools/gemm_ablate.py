#!/usr/bin/env python
"""Timing ablations of the free-running 256x256 GEMM (variant 255): which part of the K loop costs what.
mask bits: 1 = no fragment ds_reads, 2 = no LDS-DMA, 4 = no barrier/vmcnt.  Back-to-back launches (DVFS-steady)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from seed_amd import lib as L  # noqa: E402

lib = L.load()
REPS = 60
for name, M, N, K in [("square8k", 8192, 8192, 8192), ("qkv", 65792, 4224, 1408)]:
    A = torch.randn(M, K, device="cuda").bfloat16()
    W = (torch.randn(N, K, device="cuda") * 0.02).bfloat16()
    bias = torch.randn(N, device="cuda").bfloat16()
    C = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for variant, masks in ((256, [0]), (255, [int(x) for x in os.environ.get('MASKS', '0,8,12,2,7').split(',')])):
        L.check(lib.seedmi_set_option(b"gemm", variant), "opt")
        for mask in masks:
            L.check(lib.seedmi_set_option(b"gemm_ablate", mask), "opt")
            fn = lambda: L.check(lib.seedmi_gemm_bf16(M, N, K, L.ptr(A), K, L.ptr(W), K, L.ptr(bias), None, 0, L.EPI_BIAS,
                                                      L.ptr(C), N, 0, 0, L.stream_ptr()), "gemm")
            for _ in range(5):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(REPS):
                fn()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / REPS
            print(name, "variant", variant, "ablate", mask, "ms %.4f" % ms, "TF %.1f" % (2.0 * M * N * K / ms / 1e9), flush=True)
lib.seedmi_set_option(b"gemm_ablate", 0)
lib.seedmi_set_option(b"gemm", 0)
