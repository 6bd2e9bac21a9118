#!/bin/bash
# Round-2 closing run on the GPU box: the whole -m gpu suite, then every profile (tools/lab/r02_profiles.sh), then the A/B tables.
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
mkdir -p gpurun_out/r02p
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/r02p/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02p/pytest_gpu.log
timeout 2400 bash tools/lab/r02_profiles.sh > gpurun_out/r02p/profiles.log 2>&1
timeout 400 python tools/tok_ab.py "tokenize_streams=2" "tokenize_streams=2,tokenize_lnfold=0" "tokenize_streams=1" "tokenize_streams=1,tokenize_split_rounds=1" "tokenize_streams=1,tokenize_streamk=1" > gpurun_out/r02p/tok_ab.log 2>&1
cp gpurun_out/r02/tok_ab.json gpurun_out/r02p/tok_ab.json
timeout 400 python tools/gemm_sustained.py > gpurun_out/r02p/gemm_sustained.log 2>&1
cp gpurun_out/gemm_sustained.json gpurun_out/r02p/gemm_sustained.json
echo done
