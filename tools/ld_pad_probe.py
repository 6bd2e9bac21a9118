#!/usr/bin/env python
"""Does the row stride of a K-contiguous operand matter?  fc2's K = 6144 makes every row of A and W start 96 x 128 bytes apart: all rows of
a K-tile then fall on ONE of the 16 interleaved L2 channels (96 % 16 == 0), while K = 1408 (22 lines) spreads them.  Times the fc2 and
fc1 GEMMs at M = 257 (one image) and M = 65792 (B = 256) with leading dimensions K and K + PAD elements."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from seed_amd import lib as L  # noqa: E402

lib = L.load()
g = torch.Generator(device="cuda").manual_seed(0)
res = {}
for M in (257, 65792):
    for name, N, K in (("fc2", 1408, 6144), ("proj", 1408, 1408), ("fc1", 6144, 1408)):
        for pad in (0, 64, 8):
            lda = K + pad
            A = torch.randn(M, lda, device="cuda", generator=g).bfloat16()
            W = (torch.randn(N, lda, device="cuda", generator=g) * 0.02).bfloat16()
            bias = torch.zeros(N, device="cuda").bfloat16()
            C = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
            fn = lambda: L.check(lib.seedmi_gemm_bf16(M, N, K, L.ptr(A), lda, L.ptr(W), lda, L.ptr(bias), None, 0, L.EPI_BIAS, L.ptr(C), N, 0, 0,
                                                      L.stream_ptr()), "gemm")
            for _ in range(5):
                fn()
            torch.cuda.synchronize()
            ts = []
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(30):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / 30)
            ms = sorted(ts)[1]
            res[f"{name} M={M} ld=K+{pad}"] = {"us": round(ms * 1e3, 2), "tflops": round(2.0 * M * N * K / ms / 1e9, 1)}
            print(f"{name} M={M} ld=K+{pad}:", res[f"{name} M={M} ld=K+{pad}"], flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/ld_pad_probe.json", "w"), indent=1)
