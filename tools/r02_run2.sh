#!/bin/bash
# round-2, second GPU session: new kernels (GELU table, stream-K tail), new parity tests, A/B of the tokenizer options
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
mkdir -p gpurun_out/r02
timeout 900 python -m pytest tests -m gpu -x -q -s > gpurun_out/r02/pytest_run2.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02/pytest_run2.log
VARIANTS=256 REPS=40 timeout 300 python tools/gemm_sustained.py > gpurun_out/r02/gemm_sustained_run2.log 2>&1
cp gpurun_out/gemm_sustained.json gpurun_out/r02/gemm_sustained_run2.json
timeout 400 python tools/tok_ab.py > gpurun_out/r02/tok_ab.log 2>&1
timeout 600 python bench.py > gpurun_out/r02/bench_run2.json 2> gpurun_out/r02/bench_run2.err
echo done
