#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
mkdir -p gpurun_out/r02
timeout 1500 python -m pytest tests -m gpu -q -s > gpurun_out/r02/pytest_run3.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02/pytest_run3.log
timeout 600 python bench.py > gpurun_out/r02/bench_run3.json 2> gpurun_out/r02/bench_run3.err
echo done
