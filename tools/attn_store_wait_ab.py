#!/usr/bin/env python
"""ViT attention (16-wave kernel, 257 tokens, 16 heads x 88) with and without the store-tolerant K / Q wait (seedmi_set_option
"attn_store_wait"), interleaved in one process; outputs must be bit-identical."""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from seed_amd import lib as L  # noqa: E402

lib = L.load()
B, H, hd, N = int(os.environ.get("B", "128")), 16, 88, 257
C = H * hd
g = torch.Generator(device="cuda").manual_seed(0)
qkv = torch.randn(B * N, 3 * C, device="cuda", generator=g).bfloat16()
out = torch.empty(B * N, C, device="cuda", dtype=torch.bfloat16)
flops = 4.0 * B * H * N * N * hd
REPS = 20


def run():
    L.check(lib.seedmi_attention_bf16(L.ptr(qkv), 3 * C, L.ptr(qkv[:, C:]), 3 * C, L.ptr(qkv[:, 2 * C:]), 3 * C, L.ptr(out), C,
                                      B, H, hd, N, N, hd ** -0.5, 0, 1, L.stream_ptr()), "attn")


arms = [(mode, sw) for mode in (3, 4) for sw in (0, 1)]
ref, times = {}, {a: [] for a in arms}
for r in range(7):
    for a in arms:
        L.check(lib.seedmi_set_option(b"attn_vit", a[0]), "opt")
        L.check(lib.seedmi_set_option(b"attn_store_wait", a[1]), "opt")
        out.fill_(float("nan"))
        run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(REPS):
            run()
        e1.record()
        torch.cuda.synchronize()
        if r > 0:
            times[a].append(e0.elapsed_time(e1) / REPS)
        if a[0] not in ref:
            ref[a[0]] = out.clone()
        elif not torch.equal(out.view(torch.int16), ref[a[0]].view(torch.int16)):
            print(f"!! attn_vit={a[0]} attn_store_wait={a[1]}: output differs", flush=True)
for a in arms:
    med = statistics.median(times[a])
    print(f"attn_vit={a[0]} attn_store_wait={a[1]}: {med * 1e3:.1f} us  {flops / med / 1e9:.1f} TFLOP/s", flush=True)
lib.seedmi_set_option(b"attn_vit", 5)
lib.seedmi_set_option(b"attn_store_wait", 1)
