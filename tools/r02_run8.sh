#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
mkdir -p gpurun_out/r02
timeout 600 python -m pytest tests/test_batching.py tests/test_serve_contract.py -m gpu -q -x -s > gpurun_out/r02/pytest_run8a.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02/pytest_run8a.log
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r02/pytest_run8.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02/pytest_run8.log
echo done
