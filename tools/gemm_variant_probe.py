#!/usr/bin/env python
"""128x128 kernel (two 4-wave workgroups per CU, one tile per workgroup) against the 256x256 persistent kernel on the ViT shapes, plain nn.Linear
form, interleaved: how much of the 256x256 kernel's epilogue gap a second resident workgroup recovers, against what the smaller tile loses."""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from seed_amd import lib as L  # noqa: E402

lib = L.load()
B = int(os.environ.get("B", "256"))
M = B * 257
SHAPES = {"qkv": (M, 4224, 1408, L.EPI_BIAS), "proj": (M, 1408, 1408, L.EPI_BIAS), "fc1": (M, 6144, 1408, L.EPI_BIAS_GELU), "fc2": (M, 1408, 6144, L.EPI_BIAS)}
g = torch.Generator(device="cuda").manual_seed(0)
for name, (M_, N, K, epi) in SHAPES.items():
    A = torch.randn(M_, K, device="cuda", generator=g).bfloat16()
    W = (torch.randn(N, K, device="cuda", generator=g) * 0.02).bfloat16()
    bias = torch.zeros(N, device="cuda").bfloat16()
    C = torch.empty(M_, N, device="cuda", dtype=torch.bfloat16)
    VARIANTS = [int(v) for v in os.environ.get("VARIANTS", "256,128").split(",")]
    times = {v: [] for v in VARIANTS}
    ref = None
    for r in range(6):
        for v in VARIANTS:
            if 129 <= v <= 133 and epi != L.EPI_BIAS:
                continue
            L.check(lib.seedmi_set_option(b"gemm", v), "opt")
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                L.check(lib.seedmi_gemm_bf16(M_, N, K, L.ptr(A), K, L.ptr(W), K, L.ptr(bias), None, 0, epi, L.ptr(C), N, 0, 0, L.stream_ptr()), "gemm")
            e1.record()
            torch.cuda.synchronize()
            if r > 0:
                times[v].append(e0.elapsed_time(e1) / 10)
            if ref is None:
                ref = C.clone()
            elif v not in (130, 131, 132, 133) and not torch.equal(C, ref):
                print(f"!! {name}: variant {v} differs")
    fl = 2.0 * M_ * N * K
    print(name, {v: (round(statistics.median(t), 4), round(fl / statistics.median(t) / 1e9, 1)) for v, t in times.items() if t}, flush=True)
lib.seedmi_set_option(b"gemm", 0)
