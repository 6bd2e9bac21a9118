#!/usr/bin/env python
"""ViT attention kernel timing (B=256, 16 heads x 88, 257 tokens) for both V-operand variants."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from seed_amd import lib as L  # noqa: E402

lib = L.load()
B, H, hd, N = int(os.environ.get("B", "256")), 16, 88, 257
C = H * hd
g = torch.Generator(device="cuda").manual_seed(0)
qkv = torch.randn(B * N, 3 * C, device="cuda", generator=g).bfloat16()
out = torch.empty(B * N, C, device="cuda", dtype=torch.bfloat16)
flops = 4.0 * B * H * N * N * hd
for trv in (7, 6, 5, 4, 3, 2, 1):                         # 7 / 6 = staggered 16-wave kernel (flash / default), 5 / 4 / 3 = lock-step 16-wave kernel (flash + wide stores / wide stores / plain), 2 = 12-wave, 1 = attn_fullrow
    L.check(lib.seedmi_set_option(b"attn_vit", {7: 6, 6: 5, 5: 4, 4: 3, 3: 2, 2: 1}.get(trv, 0)), "opt")
    L.check(lib.seedmi_set_option(b"attn_trv", min(trv, 1)), "opt")
    ts = []
    for i in range(12):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        L.check(lib.seedmi_attention_bf16(L.ptr(qkv), 3 * C, L.ptr(qkv[:, C:]), 3 * C, L.ptr(qkv[:, 2 * C:]), 3 * C, L.ptr(out), C,
                                          B, H, hd, N, N, hd ** -0.5, 0, 1, L.stream_ptr()), "attn")
        e1.record()
        torch.cuda.synchronize()
        if i > 1:
            ts.append(e0.elapsed_time(e1))
    med = sorted(ts)[len(ts) // 2]
    print(f"trv={trv}: {med * 1e3:.1f} us  {flops / med / 1e9:.1f} TFLOP/s", flush=True)
lib.seedmi_set_option(b"attn_trv", 1)
lib.seedmi_set_option(b"attn_vit", 5)
