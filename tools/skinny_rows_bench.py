import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from seed_amd import lib as L
lib = L.load()
M = 32
g = torch.Generator(device="cuda").manual_seed(0)
for name, N, K, epi in [("qkv", 12288, 4096, L.EPI_NONE), ("o", 4096, 4096, L.EPI_BIAS_RESIDUAL), ("gate_up", 22016, 4096, L.EPI_SWIGLU), ("down", 4096, 11008, L.EPI_BIAS_RESIDUAL)]:
    ncopy = max(3, int(600e6 // (N * K * 2)) + 1)
    Wps = []
    for _ in range(ncopy):
        W = (torch.randn(N, K, device="cuda", generator=g) * 0.02).bfloat16()
        Wp = torch.empty(lib.seedmi_pack_skinny_weights_bytes(N, K) // 2, dtype=torch.bfloat16, device="cuda")
        L.check(lib.seedmi_pack_skinny_weights(L.ptr(W), K, N, K, L.ptr(Wp), L.stream_ptr()), "pack"); Wps.append(Wp); del W
    Ap = torch.randn(32 * K, device="cuda", generator=g).bfloat16()
    R = torch.randn(M, N, device="cuda", generator=g).bfloat16()
    ncol = N // 2 if epi == L.EPI_SWIGLU else N
    ldc = (ncol + 15) // 16 * 16
    C = torch.zeros(M, ldc, device="cuda", dtype=torch.bfloat16)
    line = [name]
    for rows in (1, 2, 3):
        for nw in (4, 8):
            L.check(lib.seedmi_set_option(b"skinny_rows", rows), "o"); L.check(lib.seedmi_set_option(b"skinny_waves", nw), "o")
            fn = lambda Wp: L.check(lib.seedmi_gemm_skinny_packed_bf16(M, N, K, L.ptr(Ap), K, L.ptr(Wp), L.ptr(R), N, epi, L.ptr(C), ldc, 1, 0, L.stream_ptr()), "s")
            for Wp in Wps: fn(Wp)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4):
                for Wp in Wps: fn(Wp)
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / (4 * ncopy) * 1e3
            line.append(f"R{rows}/w{nw}: {us:5.1f}us {N*K*2/us/1e6:4.2f}TB/s")
    print(" | ".join(line), flush=True)
    del Wps
