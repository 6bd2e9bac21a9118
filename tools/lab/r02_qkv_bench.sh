#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
mkdir -p gpurun_out/r02p
export TMPDIR=/tmp
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r02p/qkv -- python $R/tools/gemm_one.py 256 65792 4224 1408 200 > $R/gpurun_out/r02p/qkv.log 2>&1)
find gpurun_out/r02p/qkv -name '*kernel_stats.csv' -exec cp {} gpurun_out/r02p/qkv_gemm256_kernel_stats.csv \;
rm -rf gpurun_out/r02p/qkv
timeout 600 python bench.py > gpurun_out/r02p/bench_final.json 2> gpurun_out/r02p/bench_final.err
echo done
