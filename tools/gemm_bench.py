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
VARIANTS = [int(v) for v in os.environ.get("VARIANTS", "128,256,255").split(",")]
SHAPES = [("qkv", B * 257, 4224, 1408, L.EPI_BIAS), ("proj", B * 257, 1408, 1408, L.EPI_BIAS_RESIDUAL),
          ("fc1", B * 257, 6144, 1408, L.EPI_BIAS_GELU), ("fc2", B * 257, 1408, 6144, L.EPI_BIAS_RESIDUAL),
          ("patch", B * 256, 1408, 640, L.EPI_BIAS), ("cross_kv", B * 257, 1536, 1408, L.EPI_BIAS),
          ("qf_ffn1", B * 32, 3072, 768, L.EPI_BIAS_GELU)]
if os.environ.get("EXTRA"):                                 # epilogue-free twins and a long-K square, to separate loop from epilogue
    SHAPES = [("square8k", 8192, 8192, 8192, L.EPI_BIAS), ("proj_biasonly", B * 257, 1408, 1408, L.EPI_BIAS),
              ("fc1_biasonly", B * 257, 6144, 1408, L.EPI_BIAS), ("fc2_biasonly", B * 257, 1408, 6144, L.EPI_BIAS),
              ("square4k", 4096, 4096, 4096, L.EPI_BIAS), ("k16k", 4096, 4096, 16384, L.EPI_BIAS)]
g = torch.Generator(device="cuda").manual_seed(0)
res = {}
for name, M, N, K, epi in SHAPES:
    A = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    W = (torch.randn(N, K, device="cuda", generator=g) * 0.02).bfloat16()
    bias = (torch.randn(N, device="cuda", generator=g) * 0.1).bfloat16()
    R = torch.randn(M, N, device="cuda", generator=g).bfloat16() if epi == L.EPI_BIAS_RESIDUAL else None
    outs = {}
    times = {v: [] for v in VARIANTS}
    for rnd in range(6):
        for v in VARIANTS:                                  # persistent launch (one workgroup per CU) for all but 128
            L.check(lib.seedmi_set_option(b"gemm", v), "set_option")
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
    fl = 2.0 * M * N * K
    res[name] = {"M": M, "N": N, "K": K}
    for v in VARIANTS:
        med = sorted(times[v])[len(times[v]) // 2]
        res[name]["TF_%d" % v] = round(fl / (med * 1e-3) / 1e12, 1)
        res[name]["ms_%d" % v] = round(med, 4)
        res[name]["same_as_128_%d" % v] = torch.equal(outs[VARIANTS[0]], outs[v])
    print(name, res[name], flush=True)
lib.seedmi_set_option(b"gemm", 0)
lib.seedmi_set_option(b"gemm_persist", 1)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/gemm_bench.json", "w"), indent=1)
