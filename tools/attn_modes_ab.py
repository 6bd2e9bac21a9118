#!/usr/bin/env python
"""ViT attention (257 tokens, 16 heads x 88), lock-step 16-wave kernel (attn_vit = 3 / 4) against its staggered form (5 / 6: the two halves
of the workgroup one phase apart), interleaved in one process.  5 must equal 3 and 6 must equal 4 bit for bit - first launch and the last
of a back-to-back burst (the slots' barriers are what keeps a request from overwriting an image that is still being read)."""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from seed_amd import lib as L  # noqa: E402

lib = L.load()
H, hd, N = 16, 88, 257
C = H * hd
REPS = int(os.environ.get("REPS", "20"))
MODES = os.environ.get("MODES", "3,5,5:xcd0,4,6").split(",")       # "5:xcd0" = attn_vit 5 with the plain item walk (attn_xcd = 0)
for B in [int(v) for v in os.environ.get("BATCHES", "128").split(",")]:
    g = torch.Generator(device="cuda").manual_seed(B)
    qkv = torch.randn(B * N, 3 * C, device="cuda", generator=g).bfloat16()
    out = torch.empty(B * N, C, device="cuda", dtype=torch.bfloat16)
    flops = 4.0 * B * H * N * N * hd

    def run():
        L.check(lib.seedmi_attention_bf16(L.ptr(qkv), 3 * C, L.ptr(qkv[:, C:]), 3 * C, L.ptr(qkv[:, 2 * C:]), 3 * C, L.ptr(out), C,
                                          B, H, hd, N, N, hd ** -0.5, 0, 1, L.stream_ptr()), "attn")

    ref, times = {}, {m: [] for m in MODES}
    for r in range(int(os.environ.get("ROUNDS", "6")) + 1):
        for m in MODES:
            mode = int(m.split(":")[0])
            L.check(lib.seedmi_set_option(b"attn_vit", mode), "opt")
            L.check(lib.seedmi_set_option(b"attn_xcd", 0 if m.endswith(":xcd0") else 1), "opt")
            family = {5: 3, 6: 4}.get(mode, 3 if mode >= 7 else mode)
            for burst in (1, REPS):
                out.fill_(float("nan"))
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(burst):
                    run()
                e1.record()
                torch.cuda.synchronize()
                if family not in ref:
                    ref[family] = out.clone()
                elif not torch.equal(out.view(torch.int16), ref[family].view(torch.int16)):
                    bad = (out.view(torch.int16) != ref[family].view(torch.int16)).sum().item()
                    print(f"!! B={B} attn_vit={m} (burst {burst}): {bad} output elements differ from attn_vit={family}", flush=True)
            if r > 0:
                times[m].append(e0.elapsed_time(e1) / REPS)
    for m in MODES:
        med = statistics.median(times[m])
        print(f"B={B} attn_vit={m}: {med * 1e3:.1f} us  {flops / med / 1e9:.1f} TFLOP/s  (min {min(times[m]) * 1e3:.1f})", flush=True)
lib.seedmi_set_option(b"attn_vit", 5)
lib.seedmi_set_option(b"attn_xcd", 1)
