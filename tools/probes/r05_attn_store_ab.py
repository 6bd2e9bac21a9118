#!/usr/bin/env python
"""ViT attention (staggered 16-wave kernel, 257 tokens, 16 heads x 88): output stores as 64 bytes of 16 rows per instruction (rounds 1-4) against the
full 128-byte span of 8 rows (round 5, seedmi_set_option "attn_store" 64 | 128), interleaved in one process, at the in-path batch (128) and the
BASELINE one (256); outputs must be bit-identical (reference: the lock-step kernel, attn_vit = 3)."""
import json
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from seed_amd import lib as L  # noqa: E402

lib = L.load()
H, hd, N = 16, 88, 257
C = H * hd
REPS = 20
res = {}
for B in (128, 256):
    g = torch.Generator(device="cuda").manual_seed(0)
    qkv = torch.randn(B * N, 3 * C, device="cuda", generator=g).bfloat16()
    out = torch.empty(B * N, C, device="cuda", dtype=torch.bfloat16)

    def run():
        L.check(lib.seedmi_attention_bf16(L.ptr(qkv), 3 * C, L.ptr(qkv[:, C:]), 3 * C, L.ptr(qkv[:, 2 * C:]), 3 * C, L.ptr(out), C,
                                          B, H, hd, N, N, hd ** -0.5, 0, 1, L.stream_ptr()), "attn")

    L.check(lib.seedmi_set_option(b"attn_vit", 3), "opt")
    out.fill_(float("nan"))
    run()
    torch.cuda.synchronize()
    ref = out.clone()
    L.check(lib.seedmi_set_option(b"attn_vit", 5), "opt")
    arms = [64, 128]
    times, same = {a: [] for a in arms}, {a: True for a in arms}
    for r in range(9):
        for a in arms:
            L.check(lib.seedmi_set_option(b"attn_store", a), "opt")
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
            same[a] = same[a] and bool(torch.equal(out.view(torch.int16), ref.view(torch.int16)))
    for a in arms:
        med = statistics.median(times[a])
        res[f"B{B}_store{a}"] = {"median_us": round(med * 1e3, 2), "min_us": round(min(times[a]) * 1e3, 2), "bit_identical_to_lockstep_kernel": same[a]}
        print(f"B={B} attn_store={a}: {med * 1e3:.1f} us (min {min(times[a]) * 1e3:.1f})  bit-identical {same[a]}", flush=True)
    del qkv, out, ref
lib.seedmi_set_option(b"attn_store", 128)
out_path = os.environ.get("OUT", "gpurun_out/attn_store_ab.json")
os.makedirs(os.path.dirname(out_path) or ".", exist_ok=True)
json.dump(res, open(out_path, "w"), indent=1)
