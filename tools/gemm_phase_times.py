#!/usr/bin/env python
"""Where a K-tile's time goes inside the 256x256 GEMM (devtools build: SEEDMI_LIB_PATH=seed_amd/libseedmi_dev.so).
Waves 0 (wave group wm = 0) and 4 (wm = 1) of workgroup 0 stamp s_memtime around every barrier of eight K-tiles of their second tile:
per phase p = 1..4   p0 = LOAD section done (arrives at the first barrier) | p1 = released + fragments landed (MFMA section starts) |
                     p2 = MFMAs issued (arrives at the second barrier)    | p3 = released (next LOAD section starts)
so   LOAD = p0 - (previous p3)   wait1 = p1 - p0   MFMA = p2 - p1   wait2 = p3 - p2
plus tile-level stamps: 1 = tile opened, 90 = K loop left, 91 = next tile's requests issued, 93 = epilogue done.

    SEEDMI_LIB_PATH=seed_amd/libseedmi_dev.so SCHEDS=0,3 SHAPE=qkv python tools/gemm_phase_times.py
"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from seed_amd import lib as L  # noqa: E402

lib = L.load()
lib.seedmi_gemm_phase_timing.restype = ctypes.c_int
lib.seedmi_gemm_phase_timing.argtypes = [ctypes.c_void_p]
B = int(os.environ.get("B", "256"))
SCHEDS = [int(v) for v in os.environ.get("SCHEDS", "0").split(",")]
SHAPE = os.environ.get("SHAPE", "qkv")
M = B * 257
N, K, epi = {"qkv": (4224, 1408, L.EPI_BIAS), "fc1": (6144, 1408, L.EPI_BIAS_GELU), "proj": (1408, 1408, L.EPI_BIAS_RESIDUAL),
             "fc2": (1408, 6144, L.EPI_BIAS_RESIDUAL)}[SHAPE]
g = torch.Generator(device="cuda").manual_seed(0)
A = torch.randn(M, K, device="cuda", generator=g).bfloat16()
W = (torch.randn(N, K, device="cuda", generator=g) * 0.02).bfloat16()
C = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
if epi == L.EPI_BIAS_RESIDUAL:
    bias = (torch.randn(N, device="cuda", generator=g) * 0.1).bfloat16()
    R = torch.randn(M, N, device="cuda", generator=g).bfloat16()
    part = torch.empty(N // 64, M, 2, device="cuda")
    ext = L.GemmExt(None, None, None, L.ptr(part), M)
    args = (L.ptr(bias), L.ptr(R), N)
else:
    st = torch.rand(M + 1, 2, device="cuda", generator=g)
    cs, bf32 = torch.randn(N, device="cuda", generator=g), torch.randn(N, device="cuda", generator=g)
    ext = L.GemmExt(L.ptr(st), L.ptr(cs), L.ptr(bf32), None, 0)
    args = (None, None, 0)


def run():
    L.check(lib.seedmi_gemm_bf16_ext(M, N, K, L.ptr(A), K, L.ptr(W), K, args[0], args[1], args[2], epi, L.ptr(C), N, 0, 0, ctypes.byref(ext),
                                     None, 0, L.stream_ptr()), "gemm ext")


buf = torch.zeros(2 * 256, dtype=torch.int64, device="cuda")
for v in SCHEDS:
    L.check(lib.seedmi_set_option(b"gemm_sched", v), "gemm_sched")
    for _ in range(10):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    buf.zero_()
    L.check(lib.seedmi_gemm_phase_timing(buf.data_ptr()), "timing on")
    for _ in range(3):                                    # the last launch's stamps stay (same clocks as the timed launches)
        run()
    torch.cuda.synchronize()
    L.check(lib.seedmi_gemm_phase_timing(None), "timing off")
    print(f"== {SHAPE} sched {v}: {ms * 1e3:.1f} us per launch ({2.0 * M * N * K / ms / 1e9:.0f} TFLOP/s); stamps of the launch with timing on:")
    raw = buf.cpu().view(2, 256)
    for w in range(2):
        ev = [(int(x) >> 56 & 0xff, int(x) & ((1 << 56) - 1)) for x in raw[w].tolist() if int(x) != 0]
        if not ev:
            print(f"  wave {4 * w}: no stamps")
            continue
        # per-phase durations over the stamped K-tiles
        acc = {}
        prev_p3 = None
        tile = {}
        for i, (code, t) in enumerate(ev):
            if code in (1, 90, 91, 93):
                tile[code] = t
                if code == 1:
                    prev_p3 = t
                continue
            ph, k = divmod(code, 10)
            if k == 0 and prev_p3 is not None:
                acc.setdefault((ph, "LOAD"), []).append(t - prev_p3)
            if k == 1:
                acc.setdefault((ph, "wait1"), []).append(t - last)
            if k == 2:
                acc.setdefault((ph, "MFMA"), []).append(t - last)
            if k == 3:
                acc.setdefault((ph, "wait2"), []).append(t - last)
                prev_p3 = t
            last = t
        line = []
        total = 0.0
        for ph in (1, 2, 3, 4):
            parts = []
            for nm in ("LOAD", "wait1", "MFMA", "wait2"):
                xs = acc.get((ph, nm), [])
                xs = xs[1:] if nm == "LOAD" and ph == 1 and len(xs) > 1 else xs      # (first LOAD follows the tile-open stamp)
                m = sum(xs) / max(len(xs), 1)
                total += m
                parts.append(f"{nm} {m:5.0f}")
            line.append(f"P{ph}: " + " ".join(parts))
        print(f"  wave {4 * w}: " + " | ".join(line) + f" | K-tile {total:.0f} cycles ({len(acc.get((4, 'MFMA'), []))} stamped)")
        if 1 in tile and 90 in tile:
            print(f"          tile: open->K loop left {tile[90] - tile[1]} | ->requests issued {tile.get(91, 0) - tile[90]} | "
                  f"->epilogue done {tile.get(93, 0) - tile.get(91, 0)}   (s_memtime cycles)")
lib.seedmi_set_option(b"gemm_sched", 0)
