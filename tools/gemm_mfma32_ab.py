#!/usr/bin/env python
"""Round 4: the two-phase K-tile on v_mfma_f32_32x32x16_bf16 (devtools variant 233, gemm256t_kernel) against the shipped two-phase K-tile on
v_mfma_f32_16x16x32_bf16 (gemm256_kernel, schedule 81), interleaved in one process on the devtools library:

    SEEDMI_LIB_PATH=seed_amd/libseedmi_dev.so python tools/gemm_mfma32_ab.py

Per shape: outputs compared (bit-identical or not, max |diff|), then REPS back-to-back launches per round, ROUNDS rounds interleaved over the
arms; each arm also with the epilogue removed (gemm_ablate = 32: the K loops alone).  Plain nn.Linear epilogues (BIAS / BIAS_GELU), no fold."""
import json
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from seed_amd import lib as L  # noqa: E402

lib = L.load()
REPS = int(os.environ.get("REPS", "30"))
ROUNDS = int(os.environ.get("ROUNDS", "4"))
OUT = os.environ.get("OUT", "gpurun_out/gemm_mfma32_ab.json")
B = 256
SHAPES = [("qkv", B * 257, 4224, 1408, L.EPI_BIAS), ("fc1", B * 257, 6144, 1408, L.EPI_BIAS_GELU), ("fc2_bias", B * 257, 1408, 6144, L.EPI_BIAS),
          ("8192^3", 8192, 8192, 8192, L.EPI_BIAS)]
ARMS = [("mfma16x16x32", 256, 0), ("mfma32x32x16", 233, 0), ("mfma16x16x32_kloop", 256, 32), ("mfma32x32x16_kloop", 233, 32)]
only = [x for x in os.environ.get("SHAPES", "").split(",") if x]


def setopt(k, v):
    L.check(lib.seedmi_set_option(k.encode(), int(v)), f"{k}={v}")


res = {}
g = torch.Generator(device="cuda").manual_seed(0)
for name, M, N, K, epi in SHAPES:
    if only and name not in only:
        continue
    A = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    W = (torch.randn(N, K, device="cuda", generator=g) * 0.02).bfloat16()
    bias = (torch.randn(N, device="cuda", generator=g) * 0.1).bfloat16()
    C = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)

    def run():
        L.check(lib.seedmi_gemm_bf16(M, N, K, L.ptr(A), K, L.ptr(W), K, L.ptr(bias), None, 0, epi, L.ptr(C), N, 0, 0, L.stream_ptr()), "gemm")
    outs = {}
    for arm, variant, abl in ARMS[:2]:
        setopt("gemm", variant); setopt("gemm_ablate", abl)
        C.fill_(float("nan"))
        run()
        torch.cuda.synchronize()
        outs[arm] = C.clone()
    a, b = outs["mfma16x16x32"], outs["mfma32x32x16"]
    same = bool(torch.equal(a.view(torch.int16), b.view(torch.int16)))
    diff = (a.float() - b.float()).abs().max().item()
    finite = bool(torch.isfinite(b.float()).all())
    times = {arm: [] for arm, _, _ in ARMS}
    for r in range(ROUNDS + 1):
        for arm, variant, abl in ARMS:
            setopt("gemm", variant); setopt("gemm_ablate", abl)
            run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(REPS):
                run()
            e1.record()
            torch.cuda.synchronize()
            if r > 0:
                times[arm].append(e0.elapsed_time(e1) / REPS)
            if abl == 0:                                   # race screen on the last of the back-to-back launches
                if not torch.equal(C.view(torch.int16), outs[arm].view(torch.int16)):
                    print(f"!! {name} {arm}: launch {REPS} of round {r} differs from its first launch", flush=True)
    row = {"bit_identical": same, "max_abs_diff": diff, "finite": finite}
    for arm, _, _ in ARMS:
        med = statistics.median(times[arm])
        row[arm] = {"median_ms": round(med, 4), "min_ms": round(min(times[arm]), 4), "tflops": round(2.0 * M * N * K / med / 1e9, 1)}
    res[name] = row
    print(name, json.dumps(row), flush=True)
    del A, W, C, outs
setopt("gemm", 0); setopt("gemm_ablate", 0)
os.makedirs(os.path.dirname(OUT) or ".", exist_ok=True)
json.dump(res, open(OUT, "w"), indent=1)
