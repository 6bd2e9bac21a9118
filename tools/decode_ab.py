#!/usr/bin/env python
"""Interleaved A/B of SEED-LLaMA-8B greedy decode (B = 32, prompt 59) under different seedmi_set_option settings: one hipGraph of the
decode step is captured per arm (options are read at capture), then the arms take turns replaying STEPS steps per round, so every arm sees
the same box, clocks and cache lengths.  Prints median / min ms per step.

    python tools/decode_ab.py "" "skinny_splitk=0" "decode_attn_early=1"
"""
import json
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from seed_amd import config as C  # noqa: E402
from seed_amd import lib as L  # noqa: E402
from seed_amd.llama_engine import LlamaEngine  # noqa: E402
from seed_amd.weights import make_llama_state_dict  # noqa: E402

DEFAULTS = {"skinny_splitk": 1, "decode_attn_early": 1, "decode_fused": 1}
ARMS = [("" if a in ("-", "defaults") else a) for a in sys.argv[1:]] or ["", "skinny_splitk=0"]
STEPS = int(os.environ.get("STEPS", "16"))
ROUNDS = int(os.environ.get("ROUNDS", "5"))
OUT = os.environ.get("OUT", "gpurun_out/decode_ab.json")
lib = L.load()
cfg = C.LLAMA_8B
sd = make_llama_state_dict(cfg, seed=0, device="cuda", dtype=torch.bfloat16)
eng = LlamaEngine(sd, cfg, device="cuda", batch_cap=32, tmax=256)
del sd
g = torch.Generator(device="cuda").manual_seed(99)
prompt = torch.randint(3, 32000, (32, 59), device="cuda", generator=g)
eng.reset()
lg = eng.forward(prompt, last_only=True)
tok = lg[:, 0].float().argmax(-1, keepdim=True)
T0 = eng.past_len
n_new = 1 + (ROUNDS + 1) * STEPS


def apply(arm):
    opts = dict(DEFAULTS)
    for kv in filter(None, arm.split(",")):
        k, v = kv.split("=")
        opts[k] = int(v)
    for k, v in opts.items():
        L.check(lib.seedmi_set_option(k.encode(), v), f"{k}={v}")


graphs = []
for arm in ARMS:
    apply(arm)
    eng.past_len = T0
    replay, out = eng.capture_decode_graph(tok, n_new)
    graphs.append((arm, replay, out))
apply("")
times = {arm: [] for arm, _, _ in graphs}
for r in range(ROUNDS + 1):
    for arm, replay, _ in graphs:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        replay(STEPS)
        e1.record()
        torch.cuda.synchronize()
        if r > 0:
            times[arm].append(e0.elapsed_time(e1) / STEPS)
res = {}
ref = graphs[0][2]
for arm, _, out in graphs:
    ts = times[arm]
    res[arm or "default"] = {"median_ms": round(statistics.median(ts), 4), "min_ms": round(min(ts), 4), "all_ms": [round(t, 3) for t in ts],
                             "tokens_equal_first_arm": float((out == ref).float().mean())}
    print(f"{arm or 'default':40s} median {statistics.median(ts):.3f} ms/step  min {min(ts):.3f}  tok/s {32e3 / statistics.median(ts):.0f}  "
          f"ids equal to first arm {res[arm or 'default']['tokens_equal_first_arm']:.3f}", flush=True)
os.makedirs(os.path.dirname(OUT) or ".", exist_ok=True)
json.dump(res, open(OUT, "w"), indent=1)
