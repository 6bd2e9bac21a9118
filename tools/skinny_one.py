#!/usr/bin/env python
"""Launch the decode GEMM a few times on one shape over distinct weight copies (for rocprofv3 --pmc passes):
skinny_one.py <N> <K> [M] [swiglu]   (fragment-major weights and activations, folded RMSNorm, the split-K kernel's workspace: the launch
the decode chain issues; "swiglu" = the gate/up form with the fragment-major output)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from seed_amd import lib as L  # noqa: E402

lib = L.load()
N, K = int(sys.argv[1]), int(sys.argv[2])
M = int(sys.argv[3]) if len(sys.argv) > 3 else 32
g = torch.Generator(device="cuda").manual_seed(0)
Wps = []
for _ in range(4):                                         # > 256 MiB in rotation for the big shapes
    W = (torch.randn(N, K, device="cuda", generator=g) * 0.02).bfloat16()
    Wp = torch.empty(lib.seedmi_pack_skinny_weights_bytes(N, K) // 2, dtype=torch.bfloat16, device="cuda")
    L.check(lib.seedmi_pack_skinny_weights(L.ptr(W), K, N, K, L.ptr(Wp), L.stream_ptr()), "pack")
    Wps.append(Wp)
    del W
xp = torch.randn(((M + 15) // 16) * 16 * K, device="cuda", generator=g).bfloat16()
swiglu = len(sys.argv) > 4 and sys.argv[4] == "swiglu"
C = torch.zeros(((M + 15) // 16) * 16, N, device="cuda", dtype=torch.bfloat16)
ws = torch.zeros(lib.seedmi_gemm_skinny_workspace_bytes(), dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
for _ in range(3):
    for Wp in Wps:
        L.check(lib.seedmi_gemm_skinny_norm_ws_bf16(M, N, K, L.ptr(xp), 1, L.ptr(Wp), 1e-6, None, 0, L.EPI_SWIGLU if swiglu else L.EPI_NONE,
                                                    L.ptr(C), N // 2 if swiglu else N, 1 if swiglu else 0, None, L.ptr(ws), ws.numel(),
                                                    L.stream_ptr()), "skinny")
torch.cuda.synchronize()
