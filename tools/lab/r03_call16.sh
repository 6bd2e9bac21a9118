#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
{
echo "=== decode attention tests"; date
timeout 900 python -m pytest -q -x -m gpu "tests/test_gpu_kernels.py::test_fused_decode_attention_equals_rope_then_attention" "tests/test_gpu_kernels.py::test_skinny_gemm_with_folded_rmsnorm" 2>&1 | tail -5
echo "=== decode GEMM forms"; date
SEEDMI_LIB_PATH=seed_amd/libseedmi_dev.so ABLS=0 timeout 300 python tools/skinny_ablate.py 2>&1 | tail -8
echo "=== 8B decode, 64 steps: decode_attn_early 0 / 1 / 2, then skinny_splitk 0"; date
for o in "decode_attn_early=0" "decode_attn_early=1" "decode_attn_early=2" "decode_attn_early=0" "decode_attn_early=1" "decode_attn_early=2" "skinny_splitk=0"; do
echo -n "$o: "; DECODE_OPTS="$o" timeout 600 python tools/decode_only.py 2>&1 | tail -1
done
echo "=== llama tests (tiny + graph + batching)"; date
timeout 1200 python -m pytest -q -x -m gpu tests/test_gpu_llama.py tests/test_batching.py -k "not full_depth" 2>&1 | tail -4
date
} > gpurun_out/r03/call16.log 2>&1
tail -100 gpurun_out/r03/call16.log
