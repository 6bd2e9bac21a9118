#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd $R
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
{
echo "=== decode GEMM + attention tests"; date
timeout 900 python -m pytest -q -x -m gpu "tests/test_gpu_kernels.py::test_fused_decode_attention_equals_rope_then_attention" "tests/test_gpu_kernels.py::test_skinny_gemm_with_folded_rmsnorm" 2>&1 | tail -3
echo "=== decode GEMM forms"; date
SEEDMI_LIB_PATH=seed_amd/libseedmi_dev.so ABLS=0 timeout 300 python tools/skinny_ablate.py 2>&1 | tail -8
echo "=== decode A/B (interleaved graphs)"; date
OUT=gpurun_out/r03/decode_ab_call18.json ROUNDS=6 timeout 900 python tools/decode_ab.py "" "decode_attn_early=1" "skinny_splitk=0" "decode_attn_early=1,skinny_splitk=3" 2>&1 | tail -6
date
} > gpurun_out/r03/call18.log 2>&1
tail -60 gpurun_out/r03/call18.log
