#!/usr/bin/env python
"""Where an item's time goes in the 16-wave ViT attention kernel (devtools build: SEEDMI_LIB_PATH=seed_amd/libseedmi_dev.so).
Every wave of workgroup 0 stamps s_memtime into SGPRs at nine points of its first 8 items and stores them at the item's end:
  0 item start | 1 K, Q landed (barrier) | 2 QK^T done | 3 softmax done | 4 side job A done | 5 V landed (barrier) | 6 PV done | 7 stored |
  8 side job B done.   MODE=2|3|4 selects attn_vit (plain / 16-byte stores / + normalisation behind PV)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from seed_amd import lib as L  # noqa: E402

lib = L.load()
lib.seedmi_attn_vit_timing.restype = ctypes.c_int
lib.seedmi_attn_vit_timing.argtypes = [ctypes.c_void_p]
B, H, hd, n = int(os.environ.get("B", "128")), 16, 88, 257
D = H * hd
g = torch.Generator(device="cuda").manual_seed(0)
qkv = torch.randn(B * n, 3 * D, device="cuda", generator=g).bfloat16()
out = torch.empty(B * n, D, device="cuda", dtype=torch.bfloat16)
buf = torch.zeros(16 * 8 * 9, dtype=torch.int64, device="cuda")


def run():
    L.check(lib.seedmi_attention_bf16(L.ptr(qkv), 3 * D, qkv.data_ptr() + 2 * D, 3 * D, qkv.data_ptr() + 4 * D, 3 * D, L.ptr(out), D, B, H, hd, n, n,
                                      hd ** -0.5, 0, 1, L.stream_ptr()), "attention")


names = ["wait K", "QK^T", "softmax", "side A", "wait V", "PV", "store", "side B"]
for mode in [int(v) for v in os.environ.get("MODES", "3,4").split(",")]:
    L.check(lib.seedmi_set_option(b"attn_vit", mode), "attn_vit")
    for _ in range(20):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        run()
    e1.record()
    torch.cuda.synchronize()
    print("== attn_vit=%d: %.1f us per launch back to back" % (mode, e0.elapsed_time(e1) / 50 * 1e3))
    buf.zero_()
    L.check(lib.seedmi_attn_vit_timing(buf.data_ptr()), "timing on")
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    L.check(lib.seedmi_attn_vit_timing(None), "timing off")
    t = buf.cpu().view(16, 8, 9).double()
    for w in (0, 1, 5, 10, 15):
        d = t[w, 1:7, 1:] - t[w, 1:7, :-1]                     # items 1..6
        per = (t[w, 2:8, 0] - t[w, 1:7, 0]).mean().item()
        print("  wave %2d: " % w + "  ".join("%s %.0f" % (nm, v) for nm, v in zip(names, d.mean(0).tolist())) + "   | item period %.0f cycles" % per)
    allw = (t[:, 1:7, 1:] - t[:, 1:7, :-1]).mean(1)             # [16, 8]
    print("  mean over waves: " + "  ".join("%s %.0f" % (nm, v) for nm, v in zip(names, allw.mean(0).tolist())),
          "| max over waves: " + "  ".join("%.0f" % v for v in allw.max(0).values.tolist()))
lib.seedmi_set_option(b"attn_vit", 3)
