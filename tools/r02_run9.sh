#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
mkdir -p gpurun_out/r02
timeout 900 python -m pytest tests/test_gpu_llama.py tests/test_batching.py tests/test_gpu_kernels.py -m gpu -q -x -k "not gemm" > gpurun_out/r02/pytest_run9.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02/pytest_run9.log
bash tools/r02_profiles.sh > gpurun_out/r02/profiles_run9.log 2>&1
echo done
