#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
mkdir -p gpurun_out/r02
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_llama.py -m gpu -q -x -k "skinny or llama or decode or replay" > gpurun_out/r02/pytest_run13.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02/pytest_run13.log
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r02/prof_decode13 -- python $R/tools/decode_only.py > $R/gpurun_out/r02/decode13.log 2>&1
cd $R
f=$(ls gpurun_out/r02/prof_decode13/*/*kernel_stats.csv | head -1); head -12 $f | cut -c1-220 > gpurun_out/r02/decode13_stats.txt
rm -rf gpurun_out/r02/prof_decode13
echo done
