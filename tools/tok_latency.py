#!/usr/bin/env python
"""Latency of seedmi_tokenize at small batches (the reference scripts tokenize ONE image: scripts/seed_tokenizer_inference.py:26-29):
median of REPS host-timed encode() + synchronize per batch size, under option sets given as arguments ("" = defaults), ids compared
between the sets (options that only choose between kernels computing the same thing must not change an id).

    BATCHES=1,2,4,8 python tools/tok_latency.py "" "gemm=128"
"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from seed_amd import config as C, lib as L  # noqa: E402
from seed_amd.tokenizer_engine import TokenizerEngine  # noqa: E402
from seed_amd.weights import make_tokenizer_state_dict  # noqa: E402

lib = L.load()
BATCHES = [int(b) for b in os.environ.get("BATCHES", "1,2,4,8").split(",")]
REPS = int(os.environ.get("REPS", "30"))
sets = [("" if a in ("-", "\"\"", "defaults") else a) for a in sys.argv[1:]] or [""]
sd = make_tokenizer_state_dict(C.SEED2, seed=0, device="cuda")
eng = TokenizerEngine(sd, C.SEED2, device="cuda")
del sd
res = {}
for B in BATCHES:
    img = torch.randn(B, 3, 224, 224, device="cuda", generator=torch.Generator(device="cuda").manual_seed(5)).bfloat16()
    ref = None
    for spec in sets:
        for kv in filter(None, spec.split(",")):
            k, v = kv.split("=")
            L.check(lib.seedmi_set_option(k.encode(), int(v)), kv)
        for _ in range(3):
            ids = eng.encode(img)
        torch.cuda.synchronize()
        ts = []
        for _ in range(REPS):
            t0 = time.perf_counter()
            ids = eng.encode(img)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        ts.sort()
        same = None if ref is None else bool(torch.equal(ids, ref))
        ref = ids.clone() if ref is None else ref
        res[f"B={B} {spec or 'defaults'}"] = {"median_us": round(ts[len(ts) // 2] * 1e6, 1), "min_us": round(ts[0] * 1e6, 1),
                                              "images_per_s": round(B / ts[len(ts) // 2], 1), "ids_equal_to_first_set": same}
        for kv in filter(None, spec.split(",")):
            lib.seedmi_set_option(kv.split("=")[0].encode(), {"gemm_sched": -1, "gemm64_xcd": 1, "gemm_small": 1, "attn_small": 1, "vq_split": 1}.get(kv.split("=")[0], 0))
        print(f"B={B} {spec or 'defaults'}:", json.dumps(res[f"B={B} {spec or 'defaults'}"]), flush=True)
os.makedirs(os.path.dirname(os.environ.get("OUT", "gpurun_out/tok_latency.json")) or ".", exist_ok=True)
json.dump(res, open(os.environ.get("OUT", "gpurun_out/tok_latency.json"), "w"), indent=1)
