#!/bin/bash
# round-2 baseline at HEAD on the GPU box: tests, bench, rocprofv3 kernel stats of the bench, isolated GEMM rates, attention PMC
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
mkdir -p gpurun_out/r02
timeout 600 python -m pytest tests -m gpu -x -q > gpurun_out/r02/pytest_head.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02/pytest_head.log
timeout 600 python bench.py > gpurun_out/r02/bench_head.json 2> gpurun_out/r02/bench_head.err
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r02/prof_bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/r02/prof_bench.log 2>&1)
find gpurun_out/r02/prof_bench -name '*kernel_stats.csv' -exec cp {} gpurun_out/r02/bench_kernel_stats_head.csv \;
find gpurun_out/r02/prof_bench -type f ! -name '*kernel_stats.csv' -delete
VARIANTS=256 REPS=40 timeout 300 python tools/gemm_sustained.py > gpurun_out/r02/gemm_sustained_head.log 2>&1
cp gpurun_out/gemm_sustained.json gpurun_out/r02/gemm_sustained_head.json
B=128 timeout 120 python tools/attn_bench.py > gpurun_out/r02/attn_bench_b128.log 2>&1
timeout 400 bash tools/pmc_attn.sh > gpurun_out/r02/pmc_attn.log 2>&1
cp gpurun_out/pmc_attn_summary.json gpurun_out/r02/pmc_attn_summary_head.json
echo done
