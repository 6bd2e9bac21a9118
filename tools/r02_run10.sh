#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
mkdir -p gpurun_out/r02
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_tokenizer.py tests/test_detokenizer.py -m gpu -q -x -s -k "gemm or tokenizer or fold or batch or detok" > gpurun_out/r02/pytest_run10.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02/pytest_run10.log
timeout 400 python tools/tok_ab.py "tokenize_streams=2" "tokenize_streams=2,tokenize_lnfold=0" "tokenize_streams=1" "tokenize_streams=1,tokenize_lnfold=0" > gpurun_out/r02/tok_ab10.log 2>&1
echo done
