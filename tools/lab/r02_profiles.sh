#!/bin/bash
# Round-2 profile refresh at HEAD (run on the GPU box through gpurun; summaries are copied to profiles/ afterwards):
#   1. rocprofv3 --kernel-trace --stats of the default bench command (tokenize + decode legs)
#   2. rocprofv3 stats of the ViT QKV GEMM alone + PMC passes (traffic, L2 hit, MFMA busy, LDS conflicts) -> pmc_qkv_summary.json
#   3. rocprofv3 stats of a decode-only run (proves which kernels a graph-replayed step launches)
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
mkdir -p gpurun_out/r02p
export TMPDIR=/tmp
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r02p/bench -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/r02p/bench.log 2>&1)
find gpurun_out/r02p/bench -name '*kernel_stats.csv' -exec cp {} gpurun_out/r02p/bench_kernel_stats.csv \;
rm -rf gpurun_out/r02p/bench
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r02p/qkv -- python $R/tools/gemm_one.py 256 65792 4224 1408 200 > $R/gpurun_out/r02p/qkv.log 2>&1)
find gpurun_out/r02p/qkv -name '*kernel_stats.csv' -exec cp {} gpurun_out/r02p/qkv_gemm256_kernel_stats.csv \;
rm -rf gpurun_out/r02p/qkv
timeout 600 bash tools/pmc_qkv.sh > gpurun_out/r02p/pmc_qkv.log 2>&1
cp gpurun_out/pmc_qkv_summary.json gpurun_out/r02p/pmc_qkv_summary.json
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r02p/decode -- python $R/tools/decode_only.py > $R/gpurun_out/r02p/decode.log 2>&1)
find gpurun_out/r02p/decode -name '*kernel_stats.csv' -exec cp {} gpurun_out/r02p/decode_kernel_stats.csv \;
rm -rf gpurun_out/r02p/decode
timeout 600 python bench.py > gpurun_out/r02p/bench_final.json 2> gpurun_out/r02p/bench_final.err
echo done
