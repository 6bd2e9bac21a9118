#!/usr/bin/env python
"""Where a ViT attention item's time goes (devtools build: SEEDMI_LIB_PATH=seed_amd/libseedmi_dev.so).  Workgroup 0 stamps the cycle
counter at its phase boundaries for its first 8 items; printed per wave as cycles spent in
  wait K | QK^T + softmax | wait V / others | PV + store | end barrier
(wave 0 owns two query tiles, wave 11 one)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from seed_amd import lib as L  # noqa: E402

lib = L.load()
lib.seedmi_set_option(b"attn_vit", 1)                      # (the stamps live in the 12-wave kernel)
lib.seedmi_attn_vit_timing.restype = ctypes.c_int
lib.seedmi_attn_vit_timing.argtypes = [ctypes.c_void_p]
B, H, hd, n = int(os.environ.get("B", "128")), 16, 88, 257
D = H * hd
g = torch.Generator(device="cuda").manual_seed(0)
qkv = torch.randn(B * n, 3 * D, device="cuda", generator=g).bfloat16()
out = torch.empty(B * n, D, device="cuda", dtype=torch.bfloat16)
buf = torch.zeros(12 * 8 * 6, dtype=torch.int64, device="cuda")


def run():
    L.check(lib.seedmi_attention_bf16(L.ptr(qkv), 3 * D, qkv.data_ptr() + 2 * D, 3 * D, qkv.data_ptr() + 4 * D, 3 * D, L.ptr(out), D, B, H, hd, n, n,
                                      hd ** -0.5, 0, 1, L.stream_ptr()), "attention")


for _ in range(5):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    run()
e1.record()
torch.cuda.synchronize()
print("launch avg us: %.1f" % (e0.elapsed_time(e1) / 20 * 1e3))
L.check(lib.seedmi_attn_vit_timing(buf.data_ptr()), "timing on")
run()
torch.cuda.synchronize()
L.check(lib.seedmi_attn_vit_timing(None), "timing off")
t = buf.cpu().view(12, 8, 6)
names = ["wait K", "QK+softmax", "wait V/all", "PV+store", "end barrier"]
for w in (0, 4, 5, 11):
    d = (t[w, 1:7, 1:] - t[w, 1:7, :-1]).double()          # items 1..6, five phase lengths
    per_item = (t[w, 2:8, 0] - t[w, 1:7, 0]).double().mean().item()
    print("wave %2d: " % w + "  ".join("%s %.0f" % (nm, v) for nm, v in zip(names, d.mean(0).tolist())) + "   | item period %.0f cycles" % per_item)
