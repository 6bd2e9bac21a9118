#!/usr/bin/env python
"""Three SEED-2 tokenize passes at B = 256 and nothing else, for `rocprofv3 --kernel-trace` (tools/kernel_timeline.py +
tools/timeline_stats.py read the trace: per-kernel time, concurrency of the two sub-batch streams, idle gaps of the last pass)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from seed_amd import config as C, lib as L  # noqa: E402
from seed_amd.tokenizer_engine import TokenizerEngine  # noqa: E402
from seed_amd.weights import make_tokenizer_state_dict  # noqa: E402

lib = L.load()
for kv in (sys.argv[1].split(",") if len(sys.argv) > 1 else []):
    k, v = kv.split("=")
    L.check(lib.seedmi_set_option(k.encode(), int(v)), kv)
B = int(os.environ.get("B", "256"))
sd = make_tokenizer_state_dict(C.SEED2, seed=0, device="cuda")
eng = TokenizerEngine(sd, C.SEED2, device="cuda")
del sd
img = torch.randn(B, 3, 224, 224, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1234)).bfloat16()
for _ in range(3):
    ids = eng.encode(img)
torch.cuda.synchronize()
print(ids.shape)
