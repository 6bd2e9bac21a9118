#!/usr/bin/env python
"""Sweep the GEMM tile-order group size (L2 locality) on the ViT shapes with the persistent 256x256 kernel."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from seed_amd import lib as L
lib = L.load()
B = 256
SHAPES = [("qkv", B * 257, 4224, 1408), ("proj", B * 257, 1408, 1408), ("fc1", B * 257, 6144, 1408), ("fc2", B * 257, 1408, 6144)]
g = torch.Generator(device="cuda").manual_seed(0)
for name, M, N, K in SHAPES:
    A = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    W = (torch.randn(N, K, device="cuda", generator=g) * 0.02).bfloat16()
    bias = torch.zeros(N, device="cuda").bfloat16()
    C = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    out = [name]
    for gm in (2, 4, 6, 8, 12, 16, 32):
        L.check(lib.seedmi_set_option(b"gemm_group_m", gm), "opt")
        ts = []
        for i in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            L.check(lib.seedmi_gemm_bf16(M, N, K, L.ptr(A), K, L.ptr(W), K, L.ptr(bias), None, 0, L.EPI_BIAS, L.ptr(C), N, 0, 0, L.stream_ptr()), "g")
            e1.record(); torch.cuda.synchronize()
            if i > 1: ts.append(e0.elapsed_time(e1))
        out.append(f"gm{gm}:{2.0*M*N*K/sorted(ts)[2]/1e9:6.0f}")
    print(" ".join(out), flush=True)
