#!/usr/bin/env python
"""Calibration only: what the vendor GEMM library (hipBLASLt via torch.nn.functional.linear) reaches on the tokenizer's
shapes on this box.  Not used by the product path - it tells us how far the hand-written kernels are from the library."""
import json
import os

import torch

B = int(os.environ.get("B", "256"))
SHAPES = [("qkv", B * 257, 4224, 1408), ("proj", B * 257, 1408, 1408), ("fc1", B * 257, 6144, 1408), ("fc2", B * 257, 1408, 6144),
          ("cross_kv", B * 257, 1536, 1408), ("square8k", 8192, 8192, 8192)]
res = {}
for name, M, N, K in SHAPES:
    A = torch.randn(M, K, device="cuda").bfloat16()
    W = (torch.randn(N, K, device="cuda") * 0.02).bfloat16()
    bias = torch.randn(N, device="cuda").bfloat16()
    for _ in range(3):
        torch.nn.functional.linear(A, W, bias)
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        torch.nn.functional.linear(A, W, bias)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    med = sorted(ts)[len(ts) // 2]
    res[name] = {"M": M, "N": N, "K": K, "ms": round(med, 4), "TF": round(2.0 * M * N * K / (med * 1e-3) / 1e12, 1)}
    print(name, res[name], flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/vendor_gemm_ref.json", "w"), indent=1)
