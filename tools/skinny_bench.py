#!/usr/bin/env python
"""Decode GEMM (weight streaming) A/B on the SEED-LLaMA-8B shapes at batch 32: GB/s of weight bytes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from seed_amd import lib as L  # noqa: E402

lib = L.load()
M = int(os.environ.get("M", "32"))
SHAPES = [("qkv", 12288, 4096, L.EPI_NONE), ("o", 4096, 4096, L.EPI_BIAS_RESIDUAL), ("gate_up", 22016, 4096, L.EPI_SWIGLU),
          ("down", 4096, 11008, L.EPI_BIAS_RESIDUAL), ("lm_head", 40194, 4096, L.EPI_NONE)]
g = torch.Generator(device="cuda").manual_seed(0)
for name, N, K, epi in SHAPES:
    # 6 distinct weight copies so every timed launch streams from HBM, not from the 256 MiB MALL
    Ws = [(torch.randn(N, K, device="cuda", generator=g) * 0.02).bfloat16() for _ in range(6)]
    A = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    R = torch.randn(M, N, device="cuda", generator=g).bfloat16()
    ncol = N // 2 if epi == L.EPI_SWIGLU else N
    ldc = (ncol + 15) // 16 * 16
    C = torch.empty(M, ldc, device="cuda", dtype=torch.bfloat16)
    line = [name]
    for nt in (1, 0):
        for nw in (4, 8):
            L.check(lib.seedmi_set_option(b"skinny_nt", nt), "o")
            L.check(lib.seedmi_set_option(b"skinny_waves", nw), "o")
            ts = []
            for rnd in range(3):
                for W in Ws:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    L.check(lib.seedmi_gemm_skinny_bf16(M, N, K, L.ptr(A), K, L.ptr(W), K, L.ptr(R), N, epi, L.ptr(C), ldc,
                                                        L.stream_ptr()), "skinny")
                    e1.record()
                    torch.cuda.synchronize()
                    if rnd:
                        ts.append(e0.elapsed_time(e1))
            med = sorted(ts)[len(ts) // 2]
            line.append(f"nt{nt}/w{nw}: {med * 1e3:7.1f} us {N * K * 2 / med / 1e6:7.0f} GB/s")
    L.check(lib.seedmi_set_option(b"skinny_nt", 1), "o")
    L.check(lib.seedmi_set_option(b"skinny_waves", 0), "o")
    Wps = []
    for W in Ws:
        Wp = torch.empty(lib.seedmi_pack_skinny_weights_bytes(N, K) // 2, dtype=torch.bfloat16, device="cuda")
        L.check(lib.seedmi_pack_skinny_weights(L.ptr(W), K, N, K, L.ptr(Wp), L.stream_ptr()), "pack")
        Wps.append(Wp)
    for rows in (1, 2):
        nt = 1
        L.check(lib.seedmi_set_option(b"skinny_rows", rows), "o")
        ts = []
        for rnd in range(3):
            for Wp in Wps:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                L.check(lib.seedmi_gemm_skinny_packed_bf16(M, N, K, L.ptr(A), K, L.ptr(Wp), L.ptr(R), N, epi, L.ptr(C), ldc,
                                                           0, 0, L.stream_ptr()), "skinny_packed")
                e1.record()
                torch.cuda.synchronize()
                if rnd:
                    ts.append(e0.elapsed_time(e1))
        med = sorted(ts)[len(ts) // 2]
        line.append(f"PACKED R{rows}: {med * 1e3:7.1f} us {N * K * 2 / med / 1e6:7.0f} GB/s")
    L.check(lib.seedmi_set_option(b"skinny_rows", 0), "o")
    Ap = torch.randn(((M + 15) // 16) * 16 * K, device="cuda", generator=g).bfloat16()   # fragment-major activations (timing only)
    ts = []
    for rnd in range(3):
        for Wp in Wps:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            L.check(lib.seedmi_gemm_skinny_packed_bf16(M, N, K, L.ptr(Ap), K, L.ptr(Wp), L.ptr(R), N, epi, L.ptr(C), ldc,
                                                       1, 0, L.stream_ptr()), "skinny_packed_a")
            e1.record()
            torch.cuda.synchronize()
            if rnd:
                ts.append(e0.elapsed_time(e1))
    med = sorted(ts)[len(ts) // 2]
    line.append(f"PACKED W+A auto: {med * 1e3:7.1f} us {N * K * 2 / med / 1e6:7.0f} GB/s")
    print(" | ".join(line), flush=True)
