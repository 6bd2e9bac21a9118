#!/usr/bin/env python
"""Slot times of the staggered 16-wave ViT attention kernel (attn_vit = 5 / 6; devtools build: SEEDMI_LIB_PATH=seed_amd/libseedmi_dev.so).
Every wave of workgroup 0 stamps s_memtime at seven points of its first 8 items:
  0 slot start | 1 QK^T done | 2 barrier passed | 3 softmax (+ side scores) done | 4 barrier passed | 5 PV (+ side PV) done | 6 barrier passed
Group A = waves 0..7, group B = waves 8..15 (one slot behind)."""
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


names = ["QK^T", "wait", "softmax", "wait", "PV", "wait"]
for mode in [int(v) for v in os.environ.get("MODES", "5,6").split(",")]:
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
    t = buf.cpu().view(16, 8, 9).double()[:, :, :7]
    for w in (0, 1, 4, 7, 8, 9, 12, 15):
        d = t[w, 1:6, 1:] - t[w, 1:6, :-1]                     # items 1..5
        per = (t[w, 2:7, 0] - t[w, 1:6, 0]).mean().item()
        print("  wave %2d: " % w + "  ".join("%s %.0f" % (nm, v) for nm, v in zip(names, d.mean(0).tolist())) + "   | item period %.0f cycles" % per)
    for name, ws in (("group A", slice(0, 8)), ("group B", slice(8, 16))):
        allw = (t[ws, 1:6, 1:] - t[ws, 1:6, :-1]).mean(1)
        print("  %s mean: " % name + "  ".join("%s %.0f" % (nm, v) for nm, v in zip(names, allw.mean(0).tolist())),
              "| max: " + "  ".join("%.0f" % v for v in allw.max(0).values.tolist()))
    # slot lengths seen by wave 0 (barrier to barrier): slot 3k = t2 - t0, 3k+1 = t4 - t2, 3k+2 = t6 - t4
    w0 = t[0, 1:6]
    print("  slots (wave 0): 3k %.0f  3k+1 %.0f  3k+2 %.0f" % ((w0[:, 2] - w0[:, 0]).mean().item(), (w0[:, 4] - w0[:, 2]).mean().item(),
                                                               (w0[:, 6] - w0[:, 4]).mean().item()))
lib.seedmi_set_option(b"attn_vit", 5)
