#!/usr/bin/env python
"""Where a K-tile's time goes inside the 256x256 GEMM (devtools build: SEEDMI_LIB_PATH=seed_amd/libseedmi_dev.so).
Waves 0 (wave group wm = 0) and 4 (wm = 1) of workgroup 0 stamp s_memtime around every barrier of eight K-tiles of their second tile:
per phase p = 1..4   p0 = LOAD section done (arrives at the first barrier) | p1 = released + fragments landed (MFMA section starts) |
                     p2 = MFMAs issued (arrives at the second barrier)    | p3 = released (next LOAD section starts)
so   LOAD = p0 - (previous p3)   wait1 = p1 - p0   MFMA = p2 - p1   wait2 = p3 - p2
(every other K-tile of a window is stamped; a stamp is an s_memtime into its own SGPR pair that nothing waits for inside the stamped
K-tile), plus tile-level stamps: tile opened, K loop left, next tile's requests issued, epilogue done.

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
        # stamped K-tiles arrive as runs of codes 0..15 (4 * (phase - 1) + {0 LOAD done, 1 MFMA starts, 2 MFMAs issued, 3 released})
        tiles, tile_ev, cur = [], {}, []
        for code, t in ev:
            if code >= 100:
                tile_ev[code] = t
            else:
                if code == 0:
                    cur = []
                cur.append(t)
                if code == 16 and len(cur) == 17:
                    tiles.append(cur)
        if not tiles:
            print(f"  wave {4 * w}: no complete K-tile")
            continue
        n = len(tiles)
        avg = [sum(t[i] for t in tiles) / n for i in range(17)]
        line, total = [], 0.0
        for ph in range(4):
            b = 4 * ph
            wait1, mfma, wait2 = avg[b + 1] - avg[b], avg[b + 2] - avg[b + 1], avg[b + 3] - avg[b + 2]
            load = avg[b] - (avg[b - 1] if ph > 0 else avg[16])         # (slot 16 = the K-tile was entered, right behind the previous barrier)
            line.append(f"P{ph + 1}: LOAD {load:5.0f} wait1 {wait1:5.0f} MFMA {mfma:5.0f} wait2 {wait2:5.0f}")
        span = avg[15] - avg[16]
        print(f"  wave {4 * w}: " + " | ".join(line) + f" | K-tile {span:.0f} cycles ({n} K-tiles)")
        if 100 in tile_ev and 101 in tile_ev:
            print(f"          tile: open->K loop left {tile_ev[101] - tile_ev[100]} | ->requests issued {tile_ev.get(102, 0) - tile_ev[101]} | "
                  f"->epilogue done {tile_ev.get(103, 0) - tile_ev.get(102, 0)}   (s_memtime cycles)")
lib.seedmi_set_option(b"gemm_sched", -1)
