#!/usr/bin/env python
"""SEED-LLaMA-14B interleaved prefill (BASELINE config 5, one GPU's share: 8 sequences x 649 tokens = 4 images + 512 text)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from seed_amd import config as C
from seed_amd.llama_engine import LlamaEngine
from seed_amd.weights import make_llama_state_dict
from seed_amd import lib as L
for kv in [x for x in os.environ.get("PREFILL_OPTS", "").split(",") if x]:      # e.g. PREFILL_OPTS="gemm_group_m=4"
    k, v = kv.split("=")
    L.check(L.load().seedmi_set_option(k.encode(), int(v)), kv)
name = os.environ.get("MODEL", "14b")
cfg = C.LLAMA_14B if name == "14b" else C.LLAMA_8B
B, T = int(os.environ.get("B", "8")), 649
sd = make_llama_state_dict(cfg, seed=0, device="cuda", dtype=torch.bfloat16)
eng = LlamaEngine(sd, cfg, device="cuda", batch_cap=B, tmax=704, decode_packed=False)
del sd
g = torch.Generator(device="cuda").manual_seed(0)
ids = torch.randint(3, 32000, (B, T), device="cuda", generator=g)
for r in range(4):                                    # 4 x (<img> 32 codes </img>) interleaved with 128-token text pieces
    s = 1 + r * (128 + 34) + 128
    ids[:, s] = 32000 + 8192
    ids[:, s + 1:s + 33] = 32000 + torch.randint(0, 8192, (B, 32), device="cuda", generator=g)
    ids[:, s + 33] = 32000 + 8193
for _ in range(2):
    eng.reset(); eng.forward(ids, last_only=True)
torch.cuda.synchronize()
flops = B * T * 2.0 * cfg.linear_params() - (B * (T - 1)) * 2.0 * cfg.vocab * cfg.hidden + B * cfg.layers * 2.0 * T * T * cfg.hidden
if os.environ.get("AB"):                              # interleaved A/B of option sets: AB="prefill_streamk=0|prefill_streamk=1"
    import statistics
    arms = os.environ["AB"].split("|")
    ref, acc = None, {a: [] for a in arms}
    for r in range(int(os.environ.get("ROUNDS", "5")) + 1):
        for a in arms:
            for kv in filter(None, a.split(",")):
                k, v = kv.split("=")
                L.check(L.load().seedmi_set_option(k.encode(), int(v)), kv)
            eng.reset(); t = time.time(); lg = eng.forward(ids, last_only=True); torch.cuda.synchronize(); dt = time.time() - t
            if r > 0:
                acc[a].append(dt)
            if ref is None:
                ref = lg.clone()
            elif not torch.equal(lg, ref):
                print(f"!! {a}: logits differ from the first arm's", flush=True)
    for a in arms:
        m = statistics.median(acc[a])
        print(f"{name} prefill B={B} T={T} [{a}]: median {m * 1e3:.2f} ms  {B * T / m:.0f} tok/s  {flops / m / 1e12:.0f} TFLOP/s ({flops / m / 2.5e15:.4f} of MFMA peak)", flush=True)
    sys.exit(0)
ts = []
for _ in range(3):
    eng.reset(); t = time.time(); eng.forward(ids, last_only=True); torch.cuda.synchronize(); ts.append(time.time() - t)
dt = sorted(ts)[1]
print(f"{name} prefill B={B} T={T}: {dt * 1e3:.1f} ms  {B * T / dt:.0f} tok/s  {flops / dt / 1e12:.0f} TFLOP/s ({flops / dt / 2.5e15:.3f} of MFMA peak)", flush=True)
