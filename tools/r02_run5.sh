#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
mkdir -p gpurun_out/r02
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_tokenizer.py -m gpu -q -x > gpurun_out/r02/pytest_run5.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02/pytest_run5.log
VARIANTS=256 REPS=40 timeout 300 python tools/gemm_sustained.py > gpurun_out/r02/gemm_sustained_run5.log 2>&1
B=128 timeout 120 python tools/attn_bench.py > gpurun_out/r02/attn_bench_run5.log 2>&1
timeout 400 python tools/tok_ab.py "tokenize_streams=2" "tokenize_streams=1" > gpurun_out/r02/tok_ab5.log 2>&1
timeout 400 bash tools/pmc_attn.sh > gpurun_out/r02/pmc_attn5.log 2>&1
cp gpurun_out/pmc_attn_summary.json gpurun_out/r02/pmc_attn_summary_run5.json
echo done
