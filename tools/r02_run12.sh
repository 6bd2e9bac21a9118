#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
mkdir -p gpurun_out/r02
timeout 600 python -m pytest tests/test_gpu_tokenizer.py -m gpu -q -x > gpurun_out/r02/pytest_run12.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02/pytest_run12.log
timeout 400 python tools/tok_ab.py "tokenize_streams=2" "tokenize_streams=2,tokenize_split_rounds=0" "tokenize_streams=1" "tokenize_streams=1,tokenize_split_rounds=0" "tokenize_streams=1,tokenize_streamk=1" > gpurun_out/r02/tok_ab12.log 2>&1
echo done
