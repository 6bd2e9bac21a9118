#!/usr/bin/env python
"""Sustained (DVFS-steady) GEMM rates: each shape is launched REPS times back to back with no host sync in between, for the
hand-written kernel variants and (calibration only) the vendor library through torch.nn.functional.linear."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from seed_amd import lib as L  # noqa: E402

lib = L.load()
B = int(os.environ.get("B", "256"))
REPS = int(os.environ.get("REPS", "80"))
VARIANTS = [int(v) for v in os.environ.get("VARIANTS", "256").split(",")]
SHAPES = [("qkv", B * 257, 4224, 1408, L.EPI_BIAS), ("proj", B * 257, 1408, 1408, L.EPI_BIAS_RESIDUAL),
          ("fc1", B * 257, 6144, 1408, L.EPI_BIAS_GELU), ("fc2", B * 257, 1408, 6144, L.EPI_BIAS_RESIDUAL),
          ("square8k", 8192, 8192, 8192, L.EPI_BIAS)]
res = {}
ONLY = [x for x in os.environ.get("SHAPES", "").split(",") if x]
for name, M, N, K, epi in SHAPES:
    if ONLY and name not in ONLY:
        continue
    A = torch.randn(M, K, device="cuda").bfloat16()
    W = (torch.randn(N, K, device="cuda") * 0.02).bfloat16()
    bias = (torch.randn(N, device="cuda") * 0.1).bfloat16()
    R = torch.randn(M, N, device="cuda").bfloat16() if epi == L.EPI_BIAS_RESIDUAL else None
    C = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)

    def run(fn):
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
        return round(ms, 4), round(2.0 * M * N * K / ms / 1e9, 1)

    row = {}
    for v in VARIANTS:
        L.check(lib.seedmi_set_option(b"gemm", v), "opt")
        row["own_%d" % v] = run(lambda: L.check(lib.seedmi_gemm_bf16(
            M, N, K, L.ptr(A), K, L.ptr(W), K, L.ptr(bias), L.ptr(R), N if R is not None else 0, epi, L.ptr(C), N, 0, 0,
            L.stream_ptr()), "gemm"))
    lib.seedmi_set_option(b"gemm", 0)
    ws = torch.zeros(lib.seedmi_gemm_workspace_bytes(), dtype=torch.uint8, device="cuda")
    row["own_streamk"] = run(lambda: L.check(lib.seedmi_gemm_bf16_ws(
        M, N, K, L.ptr(A), K, L.ptr(W), K, L.ptr(bias), L.ptr(R), N if R is not None else 0, epi, L.ptr(C), N, 0, 0,
        L.ptr(ws), ws.numel(), L.stream_ptr()), "gemm ws"))
    if epi != L.EPI_BIAS:       # what the vendor line computes (bias only): the price of the fused epilogue
        row["own_streamk_bias_only"] = run(lambda: L.check(lib.seedmi_gemm_bf16_ws(
            M, N, K, L.ptr(A), K, L.ptr(W), K, L.ptr(bias), None, 0, L.EPI_BIAS, L.ptr(C), N, 0, 0,
            L.ptr(ws), ws.numel(), L.stream_ptr()), "gemm ws bias"))
    if epi == L.EPI_BIAS_RESIDUAL:
        for pf in (0, 1):       # the two residual-epilogue knobs, all four settings (library defaults: prefetch 0, nt 1)
            for nt in (0, 1):
                lib.seedmi_set_option(b"gemm_prefetch_residual", pf)
                lib.seedmi_set_option(b"gemm_residual_nt", nt)
                row["own_256_prefetch%d_nt%d" % (pf, nt)] = run(lambda: L.check(lib.seedmi_gemm_bf16(
                    M, N, K, L.ptr(A), K, L.ptr(W), K, L.ptr(bias), L.ptr(R), N, epi, L.ptr(C), N, 0, 0, L.stream_ptr()), "gemm"))
        lib.seedmi_set_option(b"gemm_prefetch_residual", 0)
        lib.seedmi_set_option(b"gemm_residual_nt", 1)
        # diagnosis: the same epilogue with an L2-resident residual (every row reads the same 2.8 KB: ldr = 0)
        row["own_256_residual_ldr0"] = run(lambda: L.check(lib.seedmi_gemm_bf16(
            M, N, K, L.ptr(A), K, L.ptr(W), K, L.ptr(bias), L.ptr(R), 0, epi, L.ptr(C), N, 0, 0, L.stream_ptr()), "gemm"))
    import ctypes
    if epi in (L.EPI_BIAS, L.EPI_BIAS_GELU) and N % 64 == 0:      # LayerNorm-fold consumer: same GEMM, fold operands by LDS-DMA
        st = torch.rand(M + 1, 2, device="cuda")
        cs, bf32 = torch.randn(N, device="cuda"), torch.randn(N, device="cuda")
        ext = L.GemmExt(L.ptr(st), L.ptr(cs), L.ptr(bf32), None, 0)
        row["own_256_lnfold"] = run(lambda: L.check(lib.seedmi_gemm_bf16_ext(
            M, N, K, L.ptr(A), K, L.ptr(W), K, None, None, 0, epi, L.ptr(C), N, 0, 0, ctypes.byref(ext), None, 0, L.stream_ptr()), "gemm ext"))
    if epi == L.EPI_BIAS_RESIDUAL and N % 64 == 0:                # ... producer: the residual epilogue also emits row statistics
        part = torch.empty(N // 64, M, 2, device="cuda")
        ext = L.GemmExt(None, None, None, L.ptr(part), M)
        row["own_256_stats"] = run(lambda: L.check(lib.seedmi_gemm_bf16_ext(
            M, N, K, L.ptr(A), K, L.ptr(W), K, L.ptr(bias), L.ptr(R), N, epi, L.ptr(C), N, 0, 0, ctypes.byref(ext), None, 0, L.stream_ptr()), "gemm ext"))
    row["vendor_linear"] = run(lambda: torch.nn.functional.linear(A, W, bias))
    res[name] = row
    print(name, row, flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/gemm_sustained.json", "w"), indent=1)
