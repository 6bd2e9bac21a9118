#!/usr/bin/env python
"""Calibration only: torch.nn.functional.linear (hipBLASLt) on one shape a few times, for rocprofv3 --pmc passes:
vendor_one.py <M> <N> <K>"""
import sys

import torch

M, N, K = (int(a) for a in sys.argv[1:4])
A = torch.randn(M, K, device="cuda").bfloat16()
W = (torch.randn(N, K, device="cuda") * 0.02).bfloat16()
b = torch.zeros(N, device="cuda").bfloat16()
for _ in range(4):
    torch.nn.functional.linear(A, W, b)
torch.cuda.synchronize()
