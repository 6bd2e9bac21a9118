#!/usr/bin/env python
"""SEED-LLaMA-8B greedy decode only (B = 32, prompt 59, 64 graph-replayed steps) for rocprofv3 --stats: which kernels a decode step
launches and how long each takes (no tokenizer, no CPU leg)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from seed_amd import config as C  # noqa: E402
from seed_amd.llama_engine import LlamaEngine  # noqa: E402
from seed_amd.weights import make_llama_state_dict  # noqa: E402

from seed_amd import lib as L  # noqa: E402
for kv in [x for x in os.environ.get("DECODE_OPTS", "").split(",") if x]:      # e.g. DECODE_OPTS="decode_persistent=1"
    k, v = kv.split("=")
    L.check(L.load().seedmi_set_option(k.encode(), int(v)), kv)
cfg = C.LLAMA_8B
sd = make_llama_state_dict(cfg, seed=0, device="cuda", dtype=torch.bfloat16)
eng = LlamaEngine(sd, cfg, device="cuda", batch_cap=32, tmax=256)
del sd
g = torch.Generator(device="cuda").manual_seed(99)
prompt = torch.randint(3, 32000, (32, 59), device="cuda", generator=g)
n_new = int(os.environ.get("N_NEW", "65"))
eng.reset()
lg = eng.forward(prompt, last_only=True)
tok = lg[:, 0].float().argmax(-1, keepdim=True)
replay, out = eng.capture_decode_graph(tok, n_new)
torch.cuda.synchronize()
t0 = time.time()
replay(n_new - 1)
torch.cuda.synchronize()
dt = time.time() - t0
print(f"decode: {(n_new - 1)} steps, {dt / (n_new - 1) * 1e3:.3f} ms/step, {32 * (n_new - 1) / dt:.0f} tok/s")
