#!/usr/bin/env python
"""Measured HBM read ceiling: seedcal_stream_read over buffers of decode-GEMM size and larger."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from seed_amd import lib as L  # noqa: E402
from tools import calib  # noqa: E402

lib = L.load()
scratch = torch.zeros(4, dtype=torch.int32, device="cuda")
res = {}
for mb in (33, 100, 180, 1024, 8192):
    buf = torch.empty(mb * 1024 * 1024, dtype=torch.uint8, device="cuda")
    buf.random_(0, 255)
    for bpc in (1, 2, 4):
        for _ in range(3):
            calib.check(calib.load().seedcal_stream_read(L.ptr(buf), buf.numel(), bpc, L.ptr(scratch), L.stream_ptr()), "stream")
        torch.cuda.synchronize()
        reps = 20 if mb < 2000 else 5
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            calib.check(calib.load().seedcal_stream_read(L.ptr(buf), buf.numel(), bpc, L.ptr(scratch), L.stream_ptr()), "stream")
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        res[f"{mb}MiB_x{bpc}"] = {"ms": round(ms, 4), "TB_s": round(buf.numel() / ms / 1e9, 3)}
        print(mb, "MiB", "blocks/CU", bpc, res[f"{mb}MiB_x{bpc}"], flush=True)
    del buf
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/hbm_read_bench.json", "w"), indent=1)
