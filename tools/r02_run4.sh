#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
mkdir -p gpurun_out/r02
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_llama.py -m gpu -q -s -k "gelu or width or gemm_residual or gemm_epilogues or streamk" > gpurun_out/r02/pytest_run4.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02/pytest_run4.log
VARIANTS=256 REPS=40 timeout 300 python tools/gemm_sustained.py > gpurun_out/r02/gemm_sustained_run4.log 2>&1
timeout 400 python tools/tok_ab.py "tokenize_streams=2" "tokenize_streams=2,gemm_prefetch_residual=0" "tokenize_streams=1" "tokenize_streams=1,gemm_prefetch_residual=0" > gpurun_out/r02/tok_ab4.log 2>&1
echo done
