#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
{
echo "=== split-K decode GEMM tests"; date
timeout 900 python -m pytest -q -x -m gpu "tests/test_gpu_kernels.py::test_skinny_gemm_with_folded_rmsnorm" 2>&1 | tail -5
echo "=== decode GEMM: split-K vs one tile per workgroup (+ ablations of the latter)"; date
SEEDMI_LIB_PATH=seed_amd/libseedmi_dev.so ABLS=0,7 timeout 300 python tools/skinny_ablate.py 2>&1 | tail -8
echo "=== 8B decode, 64 steps"; date
DECODE_OPTS="skinny_splitk=0" timeout 600 python tools/decode_only.py 2>&1 | tail -1
DECODE_OPTS="skinny_splitk=2" timeout 600 python tools/decode_only.py 2>&1 | tail -1
timeout 600 python tools/decode_only.py 2>&1 | tail -1
echo "=== llama tests"; date
timeout 1200 python -m pytest -q -x -m gpu tests/test_gpu_llama.py 2>&1 | grep -v "^$" | tail -60
date
} > gpurun_out/r03/call12.log 2>&1
tail -100 gpurun_out/r03/call12.log
