#!/usr/bin/env python
"""Where a decode GEMM launch loses time against a plain stream of the same bytes (devtools library: SEEDMI_LIB_PATH=seed_amd/libseedmi_dev.so).
seedmi_set_option("skinny_ablate", bits): 1 = no activation loads, 2 = no MFMA / norm sums, 4 = no cross-wave reduction / epilogue.
Every shape is issued as the decode chain issues it (fragment-major W and A, folded RMSNorm where the chain folds it); weights rotate over
> 600 MB of copies so every launch streams from HBM.  The last column is seedcal_stream_read over the same number of bytes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from seed_amd import lib as L  # noqa: E402
from tools import calib  # noqa: E402

lib = L.load()
M = int(os.environ.get("M", "32"))
ABLS = [int(v) for v in os.environ.get("ABLS", "0,1,2,4,3,7").split(",")]
g = torch.Generator(device="cuda").manual_seed(0)
scratch = torch.zeros(4, dtype=torch.int32, device="cuda")
for name, N, K, epi, eps in [("qkv", 12288, 4096, L.EPI_NONE, 1e-6), ("o", 4096, 4096, L.EPI_BIAS_RESIDUAL, 0.0),
                             ("gate_up", 22016, 4096, L.EPI_SWIGLU, 1e-6), ("down", 4096, 11008, L.EPI_BIAS_RESIDUAL, 0.0)]:
    ncopy = max(3, int(600e6 // (N * K * 2)) + 1)
    Wps = []
    for _ in range(ncopy):
        W = (torch.randn(N, K, device="cuda", generator=g) * 0.02).bfloat16()
        Wp = torch.empty(lib.seedmi_pack_skinny_weights_bytes(N, K) // 2, dtype=torch.bfloat16, device="cuda")
        L.check(lib.seedmi_pack_skinny_weights(L.ptr(W), K, N, K, L.ptr(Wp), L.stream_ptr()), "pack")
        Wps.append(Wp)
        del W
    Ap = torch.randn(32 * K, device="cuda", generator=g).bfloat16()
    R = torch.randn(M, N, device="cuda", generator=g).bfloat16()
    ncol = N // 2 if epi == L.EPI_SWIGLU else N
    C = torch.zeros(M * ncol + 64, device="cuda", dtype=torch.bfloat16)
    Xp = torch.zeros(32 * N, device="cuda", dtype=torch.bfloat16)
    line = [name]

    def run(Wp):
        res = L.ptr(R) if epi == L.EPI_BIAS_RESIDUAL else None
        xp = L.ptr(Xp) if epi == L.EPI_BIAS_RESIDUAL else None
        L.check(lib.seedmi_gemm_skinny_norm_bf16(M, N, K, L.ptr(Ap), 1, L.ptr(Wp), eps, res, N, epi, L.ptr(C), ncol,
                                                 1 if epi == L.EPI_SWIGLU else 0, xp, L.stream_ptr()), "skinny")

    def timed(fn):
        for Wp in Wps:
            fn(Wp)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4):
            for Wp in Wps:
                fn(Wp)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / (4 * ncopy) * 1e3

    ws = torch.zeros(lib.seedmi_gemm_skinny_workspace_bytes(), dtype=torch.uint8, device="cuda")

    def run_sk(Wp):
        res = L.ptr(R) if epi == L.EPI_BIAS_RESIDUAL else None
        xp = L.ptr(Xp) if epi == L.EPI_BIAS_RESIDUAL else None
        L.check(lib.seedmi_gemm_skinny_norm_ws_bf16(M, N, K, L.ptr(Ap), 1, L.ptr(Wp), eps, res, N, epi, L.ptr(C), ncol,
                                                    1 if epi == L.EPI_SWIGLU else 0, xp, L.ptr(ws), ws.numel(), L.stream_ptr()), "skinny sk")

    for mode, tag in ((1, "split-K kernel, by shape"), (2, "split-K kernel, cut"), (4, "cut, 2 workgroups per CU")):
        L.check(lib.seedmi_set_option(b"skinny_splitk", mode), "skinny_splitk")
        us = timed(run_sk)
        line.append(f"{tag}: {us:5.1f}us {N * K * 2 / us / 1e6:4.2f}TB/s")
    L.check(lib.seedmi_set_option(b"skinny_splitk", 1), "skinny_splitk")
    assert int(ws[:4096].view(torch.int32).abs().sum()) == 0, "flag words"
    for abl in ABLS:
        L.check(lib.seedmi_set_option(b"skinny_ablate", abl), "skinny_ablate")
        us = timed(run)
        line.append(f"abl{abl}: {us:5.1f}us {N * K * 2 / us / 1e6:4.2f}TB/s")
    L.check(lib.seedmi_set_option(b"skinny_ablate", 0), "skinny_ablate")
    nbytes = N * K * 2
    us = timed(lambda Wp: calib.check(calib.load().seedcal_stream_read(L.ptr(Wp), nbytes, 1, L.ptr(scratch), L.stream_ptr()), "stream"))
    line.append(f"stream: {us:5.1f}us {nbytes / us / 1e6:4.2f}TB/s")
    print(" | ".join(line), flush=True)
    del Wps
