#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
mkdir -p gpurun_out/r02
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "gemm" > gpurun_out/r02/pytest_run16.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02/pytest_run16.log
for i in 1 2; do
SHAPES=proj,fc2 timeout 300 python tools/gemm_sustained.py > gpurun_out/r02/sustained16_new$i.log 2>&1
SEEDMI_LIB_PATH=$R/seed_amd/_build_ab/lib_prev.so SHAPES=proj,fc2 timeout 300 python tools/gemm_sustained.py > gpurun_out/r02/sustained16_prev$i.log 2>&1
done
timeout 300 python tools/tok_ab.py "tokenize_streams=2" "tokenize_streams=1" > gpurun_out/r02/tok_ab16_new.log 2>&1
SEEDMI_LIB_PATH=$R/seed_amd/_build_ab/lib_prev.so timeout 300 python tools/tok_ab.py "tokenize_streams=2" "tokenize_streams=1" > gpurun_out/r02/tok_ab16_prev.log 2>&1
echo done
