#!/usr/bin/env python
"""Tokenizer throughput / latency vs batch size (full SEED-2 model, synthetic weights)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from seed_amd import config as C
from seed_amd.tokenizer_engine import TokenizerEngine
from seed_amd.weights import make_tokenizer_state_dict
sd = make_tokenizer_state_dict(C.SEED2, seed=0, device="cuda")
eng = TokenizerEngine(sd, C.SEED2)
del sd
for B in (1, 2, 8, 32, 64, 128, 256, 512, 1024):
    x = torch.randn(B, 3, 224, 224, device="cuda").bfloat16()
    for _ in range(2):
        eng.encode(x)
    torch.cuda.synchronize()
    n = 5 if B >= 128 else 20
    t = time.time()
    for _ in range(n):
        eng.encode(x)
    torch.cuda.synchronize()
    dt = (time.time() - t) / n
    print(f"B={B:5d}  {dt * 1e3:9.3f} ms/batch  {B / dt:9.1f} img/s", flush=True)
