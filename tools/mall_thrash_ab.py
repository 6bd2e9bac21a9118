#!/usr/bin/env python
"""Is the ViT QKV GEMM's 2.5x-algorithmic traffic past the L2s served by the Infinity Cache, and would it matter if it were not?
("HBM sees ~1x" was an argument, there being no HBM-side counter that separates Infinity-Cache hits - TCC_EA0_RDREQ_DRAM
counts every memory-side read, profiles/r06_pmc_stalls_qkv_gemm256.json.)  A/B on the launch itself: the GEMM back to back (A = 185 MB and
W = 12 MB stay in the 256 MB Infinity Cache between launches) against the same launch behind a 1 GiB stream read that evicts them (every
operand byte of the launch then comes from HBM at least once, re-reads within the launch may still hit).  Each launch timed by its own pair of
HIP events.  Equal times = the operand stream does not bound the launch whichever memory serves it."""
import ctypes
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from seed_amd import lib as L  # noqa: E402
from tools import calib  # noqa: E402

lib, cal = L.load(), calib.load()
M, N, K = 65792, 4224, 1408
g = torch.Generator(device="cuda").manual_seed(1)
A = torch.randn(M, K, device="cuda", generator=g).bfloat16()
W = (torch.randn(N, K, device="cuda", generator=g) * 0.02).bfloat16()
Cc = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
stats = torch.zeros(M + (M & 1), 2, dtype=torch.float32, device="cuda")
L.check(lib.seedmi_layernorm_stats_bf16(L.ptr(A), K, M, K, 1e-6, L.ptr(stats), L.stream_ptr()), "stats")
cs, b32 = W.float().sum(1).contiguous(), torch.zeros(N, device="cuda")
ext = L.GemmExt(L.ptr(stats), L.ptr(cs), L.ptr(b32), None, 0)
junk = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
junk.view(torch.int32).fill_(0x01020304)
scratch = torch.zeros(4, dtype=torch.float32, device="cuda")


def gemm():
    L.check(lib.seedmi_gemm_bf16_ext(M, N, K, L.ptr(A), K, L.ptr(W), K, None, None, 0, L.EPI_BIAS, L.ptr(Cc), N, 0, 0, ctypes.byref(ext), None, 0,
                                     L.stream_ptr()), "gemm")


def thrash():
    calib.check(cal.seedcal_stream_read(L.ptr(junk), junk.numel(), 4, L.ptr(scratch), L.stream_ptr()), "stream read")


def timed(evict, n=20):
    for _ in range(3):
        if evict:
            thrash()
        gemm()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        if evict:
            thrash()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        gemm()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2], ts[0]


res = {}
for rnd in range(3):                                    # interleaved rounds (clock drift, DVFS)
    for name, ev in (("resident", False), ("evicted", True)):
        med, mn = timed(ev)
        res.setdefault(name, []).append(round(med, 4))
flops = 2.0 * M * N * K
out = {"launch": "ViT QKV GEMM, M=65792 K=1408 N=4224, LayerNorm fold, BIAS", "median_ms_operands_resident_in_infinity_cache": res["resident"],
       "median_ms_behind_a_1GiB_eviction_sweep": res["evicted"],
       "tflops_resident": round(flops / (sorted(res["resident"])[1] * 1e-3) / 1e12, 1), "tflops_evicted": round(flops / (sorted(res["evicted"])[1] * 1e-3) / 1e12, 1),
       "note": "single launches between their own events (a lone launch runs on a cooler clock than bench.py's bursts: compare the two arms, not with the bench line)"}
print(json.dumps(out, indent=1))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/mall_thrash_ab.json", "w"), indent=1)
