#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
mkdir -p gpurun_out/r02
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "gemm" > gpurun_out/r02/pytest_run7.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02/pytest_run7.log
VARIANTS=256,257 REPS=40 timeout 400 python tools/gemm_sustained.py > gpurun_out/r02/gemm_sustained_run7.log 2>&1
timeout 400 python tools/tok_ab.py "tokenize_streams=2" "tokenize_streams=2,gemm=257" "tokenize_streams=1,gemm=257" > gpurun_out/r02/tok_ab7.log 2>&1
echo done
