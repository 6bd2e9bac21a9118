#!/usr/bin/env python
"""The four ViT GEMMs as ONE image issues them (M = 257: the 64x64 small-M kernel), timed in bursts between two HIP events:
hot (the same weights every launch: L2 / Infinity-Cache resident after the first) against cold (a ring of weight copies larger than the
256 MB Infinity Cache: every launch streams its weights from HBM, as the one-image tokenize pass does), plain nn.Linear epilogues and
the in-path forms (LayerNorm fold consumed from span planes / residual + statistics), and - with the devtools build - without the
epilogue (gemm_ablate 32) and without its stores (33).

    M=257 python tools/gemm_small_bench.py            SEEDMI_LIB_PATH=seed_amd/libseedmi_dev.so ABLATE=0,32,33 python tools/gemm_small_bench.py
"""
import ctypes
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from seed_amd import lib as L  # noqa: E402

lib = L.load()
M = int(os.environ.get("M", "257"))
ABL = [int(a) for a in os.environ.get("ABLATE", "0").split(",")]
OPTS = os.environ.get("OPTS", "")
for kv in filter(None, OPTS.split(",")):
    k, v = kv.split("=")
    L.check(lib.seedmi_set_option(k.encode(), int(v)), kv)
D, F = 1408, 6144
SHAPES = [("qkv", 3 * D, D, L.EPI_BIAS, "fold"), ("proj", D, D, L.EPI_BIAS_RESIDUAL, "stats"), ("fc1", F, D, L.EPI_BIAS_GELU, "fold"),
          ("fc2", D, F, L.EPI_BIAS_RESIDUAL, "stats")]
g = torch.Generator(device="cuda").manual_seed(0)
res = {}
for name, N, K, epi, role in SHAPES:
    ncopies = max(2, int(320e6 // (N * K * 2)) + 1)                 # > 256 MB of weights in rotation = cold
    Ws = [(torch.randn(N, K, device="cuda", generator=g) * 0.02).bfloat16() for _ in range(ncopies)]
    A = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    bias = (torch.randn(N, device="cuda", generator=g) * 0.1).bfloat16()
    R = torch.randn(M, N, device="cuda", generator=g).bfloat16()
    C = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    spans = (D + 63) // 64
    planes = torch.rand(spans, M, 2, device="cuda", generator=g) * 64          # (sum, sum of squares) partials of a 1408-wide row
    planes[..., 1] += 64.0
    cs, b32 = Ws[0].float().sum(1).contiguous(), bias.float().contiguous()
    spart = torch.zeros((N + 63) // 64, M, 2, device="cuda")

    def launch(W, form):
        if form == "plain":
            return lib.seedmi_gemm_bf16(M, N, K, L.ptr(A), K, L.ptr(W), K, L.ptr(bias), L.ptr(R) if epi == L.EPI_BIAS_RESIDUAL else None,
                                        N if epi == L.EPI_BIAS_RESIDUAL else 0, epi, L.ptr(C), N, 0, 0, L.stream_ptr())
        if role == "fold":
            ext = L.GemmExt(L.ptr(planes), L.ptr(cs), L.ptr(b32), None, 0, 0, spans, M, D, 1e-6)
            return lib.seedmi_gemm_bf16_ext(M, N, K, L.ptr(A), K, L.ptr(W), K, None, None, 0, epi, L.ptr(C), N, 0, 0, ctypes.byref(ext), None, 0,
                                            L.stream_ptr())
        ext = L.GemmExt(None, None, None, L.ptr(spart), M)
        return lib.seedmi_gemm_bf16_ext(M, N, K, L.ptr(A), K, L.ptr(W), K, L.ptr(bias), L.ptr(R), N, epi, L.ptr(C), N, 0, 0, ctypes.byref(ext), None, 0,
                                        L.stream_ptr())

    def burst(form, cold, n=40):
        for i in range(4):
            L.check(launch(Ws[i % ncopies] if cold else Ws[0], form), "gemm")
        torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(n):
                launch(Ws[i % ncopies] if cold else Ws[0], form)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / n * 1e3)
        return round(sorted(ts)[1], 2)
    row = {"N": N, "K": K, "weights_MB": round(N * K * 2 / 1e6, 1)}
    for abl in ABL:
        if abl:
            L.check(lib.seedmi_set_option(b"gemm_ablate", abl), "ablate (needs the devtools build)")
        tag = "" if not abl else f"_ablate{abl}"
        for form in ("plain", "in_path"):
            row[f"{form}_hot_us{tag}"] = burst(form, False)
            row[f"{form}_cold_us{tag}"] = burst(form, True)
        if abl:
            lib.seedmi_set_option(b"gemm_ablate", 0)
    row["cold_in_path_GBps"] = round(N * K * 2 / row["in_path_cold_us"] / 1e3, 1)
    res[name] = row
    print(name, json.dumps(row), flush=True)
    del Ws
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open(os.environ.get("OUT", "gpurun_out/gemm_small_bench.json"), "w"), indent=1)
