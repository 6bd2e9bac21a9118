#!/usr/bin/env python
"""Sweep the GEMM tile-order group size (m-tiles per L2 tile group) on the ViT shapes with the persistent 256x256 kernel: back-to-back
launches, the settings visited twice in alternating order (clock drift shows up as a difference between the two visits)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from seed_amd import lib as L  # noqa: E402

lib = L.load()
B = 256
REPS = 40
SHAPES = [("qkv", B * 257, 4224, 1408, L.EPI_BIAS), ("proj", B * 257, 1408, 1408, L.EPI_BIAS_RESIDUAL), ("fc1", B * 257, 6144, 1408, L.EPI_BIAS_GELU),
          ("fc2", B * 257, 1408, 6144, L.EPI_BIAS_RESIDUAL)]
GMS = (2, 3, 4, 6, 8, 12, 16)
g = torch.Generator(device="cuda").manual_seed(0)
for name, M, N, K, epi in SHAPES:
    A = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    W = (torch.randn(N, K, device="cuda", generator=g) * 0.02).bfloat16()
    bias = torch.zeros(N, device="cuda").bfloat16()
    R = torch.randn(M, N, device="cuda", generator=g).bfloat16() if epi == L.EPI_BIAS_RESIDUAL else None
    C = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)

    def run():
        L.check(lib.seedmi_gemm_bf16(M, N, K, L.ptr(A), K, L.ptr(W), K, L.ptr(bias), L.ptr(R), N if R is not None else 0, epi, L.ptr(C), N, 0, 0,
                                     L.stream_ptr()), "gemm")
    res = {gm: [] for gm in GMS}
    for order in (GMS, GMS[::-1]):
        for gm in order:
            L.check(lib.seedmi_set_option(b"gemm_group_m", gm), "opt")
            for _ in range(5):
                run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(REPS):
                run()
            e1.record()
            torch.cuda.synchronize()
            res[gm].append(2.0 * M * N * K / (e0.elapsed_time(e1) / REPS) / 1e9)
    print(name, " ".join("gm%d:%.0f/%.0f" % (gm, v[0], v[1]) for gm, v in res.items()), flush=True)
lib.seedmi_set_option(b"gemm_group_m", 0)
