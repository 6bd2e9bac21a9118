#!/usr/bin/env python
"""GEMM kernel A/B on the tokenizer's shapes (interleaved rounds in one process, HIP-event timed).
Also checks that the 256x256 kernel is BIT-identical to the 128x128 kernel (same fp32 accumulation order)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from seed_amd import lib as L  # noqa: E402

lib = L.load()
B = int(os.environ.get("B", "256"))
SHAPES = [("qkv", B * 257, 4224, 1408, L.EPI_BIAS), ("proj", B * 257, 1408, 1408, L.EPI_BIAS_RESIDUAL),
          ("fc1", B * 257, 6144, 1408, L.EPI_BIAS_GELU), ("fc2", B * 257, 1408, 6144, L.EPI_BIAS_RESIDUAL),
          ("patch", B * 256, 1408, 640, L.EPI_BIAS), ("cross_kv", B * 257, 1536, 1408, L.EPI_BIAS),
          ("qf_ffn1", B * 32, 3072, 768, L.EPI_BIAS_GELU)]
g = torch.Generator(device="cuda").manual_seed(0)
res = {}
for name, M, N, K, epi in SHAPES:
    A = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    W = (torch.randn(N, K, device="cuda", generator=g) * 0.02).bfloat16()
    bias = (torch.randn(N, device="cuda", generator=g) * 0.1).bfloat16()
    R = torch.randn(M, N, device="cuda", generator=g).bfloat16() if epi == L.EPI_BIAS_RESIDUAL else None
    outs = {}
    times = {128: [], 256: [], 257: [], 232: []}
    for rnd in range(6):
        for v in (128, 256, 257, 232):                      # 257 = the 256 kernel launched persistent (one workgroup per CU)
            L.check(lib.seedmi_set_option(b"gemm", 256 if v == 257 else v), "set_option")
            L.check(lib.seedmi_set_option(b"gemm_persist", 1 if v in (257, 232) else 0), "set_option")
            C = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            L.check(lib.seedmi_gemm_bf16(M, N, K, L.ptr(A), K, L.ptr(W), K, L.ptr(bias), L.ptr(R), N if R is not None else 0,
                                         epi, L.ptr(C), N, 0, 0, L.stream_ptr()), "gemm")
            e1.record()
            torch.cuda.synchronize()
            if rnd > 0:
                times[v].append(e0.elapsed_time(e1))
            outs[v] = C
    same = torch.equal(outs[128], outs[256]) and torch.equal(outs[128], outs[257])
    same32 = torch.equal(outs[128], outs[232])
    relx = ((outs[232].float() - outs[128].float()).norm() / outs[128].float().norm()).item()
    fl = 2.0 * M * N * K
    r = {v: round(fl / (sorted(t)[len(t) // 2] * 1e-3) / 1e12, 1) for v, t in times.items()}
    res[name] = {"M": M, "N": N, "K": K, "TF_128": r[128], "TF_256": r[256], "TF_256_persistent": r[257], "TF_256x32": r[232], "bit_identical": same, "x32_identical": same32, "x32_rel": relx,
                 "ms_128": round(sorted(times[128])[2], 4), "ms_256": round(sorted(times[256])[2], 4)}
    print(name, res[name], flush=True)
lib.seedmi_set_option(b"gemm", 0)
lib.seedmi_set_option(b"gemm_persist", 1)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/gemm_bench.json", "w"), indent=1)
