#!/usr/bin/env python
"""Upper bound of what a weight prefetch could return to a decode launch: bench.py's per-kernel bursts (8B, B = 32) with a different
layer's weights per launch (cold: every byte from HBM, as in the step) against the SAME layer's weights every launch (resident in the
Infinity Cache, partly in L2)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from seed_amd import config as C  # noqa: E402
from seed_amd.llama_engine import LlamaEngine  # noqa: E402
from seed_amd.weights import make_llama_state_dict  # noqa: E402

cfg = C.LLAMA_8B
sd = make_llama_state_dict(cfg, seed=0, device="cuda", dtype=torch.bfloat16)
eng = LlamaEngine(sd, cfg, device="cuda", batch_cap=32, tmax=256)
del sd
cold = bench.decode_per_kernel(eng, cfg, 32, 123)
layers = eng._layers
hot_layers = (type(layers[0]) * cfg.layers)(*[layers[0]] * cfg.layers)
eng._layers = hot_layers
hot = bench.decode_per_kernel(eng, cfg, 32, 123)
out = {k: {"cold_us": cold["launches"][k]["us"], "hot_us": hot["launches"][k]["us"], "MB": round(cold["launches"][k]["bytes"] / 1e6, 1)} for k in cold["launches"]}
out["layer_sum_us"] = {"cold": cold["layer_sum_us"], "hot": hot["layer_sum_us"]}
print(json.dumps(out, indent=1))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/decode_hot_cold.json", "w"), indent=1)
