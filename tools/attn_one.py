#!/usr/bin/env python
"""A few launches of the ViT attention (257 tokens, 16 heads x 88) and nothing else, for rocprofv3 --pmc passes:
    python tools/attn_one.py <batch> <attn_vit> <attn_xcd> [launches]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from seed_amd import lib as L  # noqa: E402

lib = L.load()
B, mode, xcd = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
n_launch = int(sys.argv[4]) if len(sys.argv) > 4 else 3
H, hd, N = 16, 88, 257
C = H * hd
L.check(lib.seedmi_set_option(b"attn_vit", mode), "attn_vit")
L.check(lib.seedmi_set_option(b"attn_xcd", xcd), "attn_xcd")
g = torch.Generator(device="cuda").manual_seed(0)
qkv = torch.randn(B * N, 3 * C, device="cuda", generator=g).bfloat16()
out = torch.empty(B * N, C, device="cuda", dtype=torch.bfloat16)
junk = torch.empty(1 << 28, dtype=torch.uint8, device="cuda")          # 256 MiB: pushes qkv out of the Infinity Cache between launches
for _ in range(n_launch):
    junk.fill_(1)
    L.check(lib.seedmi_attention_bf16(L.ptr(qkv), 3 * C, L.ptr(qkv[:, C:]), 3 * C, L.ptr(qkv[:, 2 * C:]), 3 * C, L.ptr(out), C,
                                      B, H, hd, N, N, hd ** -0.5, 0, 1, L.stream_ptr()), "attn")
torch.cuda.synchronize()
