#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd $R
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
{
echo "=== attention tests (buffer-form staging, one lane offset per call: pieces p0 + 3 j)"; date
timeout 600 python -m pytest -q -m gpu "tests/test_gpu_kernels.py::test_attention_fullrow" tests/test_gpu_tokenizer.py 2>&1 | tail -3
echo "=== attention bench B=128 / 256"; date
B=128 timeout 300 python tools/attn_bench.py 2>&1 | tail -6
B=256 timeout 300 python tools/attn_bench.py 2>&1 | tail -6
echo "=== end-to-end"; date
ROUNDS=4 OUT=gpurun_out/r03/tok_ab_call22.json timeout 600 python tools/tok_ab.py "" "attn_vit=1" 2>&1 | tail -12
date
} > gpurun_out/r03/call22.log 2>&1
tail -40 gpurun_out/r03/call22.log
