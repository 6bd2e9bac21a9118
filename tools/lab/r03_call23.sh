#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd $R
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
{
echo "=== decode GEMM per shape: weight loads non-temporal (shipped) vs plain (variant build)"; date
for lib in seed_amd/libseedmi.so seed_amd/libseedmi_sknt0.so seed_amd/libseedmi.so seed_amd/libseedmi_sknt0.so; do
echo "--- $lib"
SEEDMI_LIB_PATH=$lib ABLS= timeout 300 python - <<'PY'
import os, sys, torch
sys.path.insert(0, os.getcwd())
from seed_amd import lib as L
lib = L.load()
M = 32
g = torch.Generator(device="cuda").manual_seed(0)
for name, N, K, epi, eps in [("qkv", 12288, 4096, L.EPI_NONE, 1e-6), ("o", 4096, 4096, L.EPI_BIAS_RESIDUAL, 0.0), ("gate_up", 22016, 4096, L.EPI_SWIGLU, 1e-6), ("down", 4096, 11008, L.EPI_BIAS_RESIDUAL, 0.0)]:
    ncopy = max(3, int(600e6 // (N * K * 2)) + 1)
    Wps = []
    for _ in range(ncopy):
        W = (torch.randn(N, K, device="cuda", generator=g) * 0.02).bfloat16()
        Wp = torch.empty(lib.seedmi_pack_skinny_weights_bytes(N, K) // 2, dtype=torch.bfloat16, device="cuda")
        L.check(lib.seedmi_pack_skinny_weights(L.ptr(W), K, N, K, L.ptr(Wp), L.stream_ptr()), "pack"); Wps.append(Wp); del W
    Ap = torch.randn(32 * K, device="cuda", generator=g).bfloat16()
    Rr = torch.randn(M, N, device="cuda", generator=g).bfloat16()
    ncol = N // 2 if epi == L.EPI_SWIGLU else N
    C = torch.zeros(M * ncol + 64, device="cuda", dtype=torch.bfloat16)
    Xp = torch.zeros(32 * N, device="cuda", dtype=torch.bfloat16)
    ws = torch.zeros(lib.seedmi_gemm_skinny_workspace_bytes(), dtype=torch.uint8, device="cuda")
    def run(Wp):
        res = L.ptr(Rr) if epi == L.EPI_BIAS_RESIDUAL else None
        xp = L.ptr(Xp) if epi == L.EPI_BIAS_RESIDUAL else None
        L.check(lib.seedmi_gemm_skinny_norm_ws_bf16(M, N, K, L.ptr(Ap), 1, L.ptr(Wp), eps, res, N, epi, L.ptr(C), ncol, 1 if epi == L.EPI_SWIGLU else 0, xp, L.ptr(ws), ws.numel(), L.stream_ptr()), "sk")
    for Wp in Wps: run(Wp)
    torch.cuda.synchronize()
    ts = []
    for rep in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4):
            for Wp in Wps: run(Wp)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / (4 * ncopy) * 1e3)
    ts.sort()
    print(f"{name}: {ts[2]:.1f} us {N*K*2/ts[2]/1e6:.2f} TB/s", flush=True)
    del Wps
PY
done
echo "=== 8B decode step"; date
for lib in seed_amd/libseedmi.so seed_amd/libseedmi_sknt0.so seed_amd/libseedmi.so seed_amd/libseedmi_sknt0.so; do
echo -n "$lib: "; SEEDMI_LIB_PATH=$lib timeout 600 python tools/decode_only.py 2>&1 | tail -1
done
date
} > gpurun_out/r03/call23.log 2>&1
tail -50 gpurun_out/r03/call23.log
