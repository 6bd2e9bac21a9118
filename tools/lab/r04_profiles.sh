#!/bin/bash
# Round-4 profile refresh at HEAD (run on the GPU box through gpurun; summaries are copied to profiles/ afterwards):
#   1. rocprofv3 --kernel-trace --stats of the default bench command (tokenize + decode legs)
#   2. rocprofv3 stats of the ViT QKV GEMM alone + PMC passes (traffic, L2 hit, MFMA busy, LDS conflicts) -> pmc_qkv_summary.json
#   3. rocprofv3 stats of a decode-only run + PMC passes of the q/k/v and gate/up decode GEMMs
#   4. the default bench line
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
O=gpurun_out/r04p
mkdir -p $O
export TMPDIR=/tmp
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/bench -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $R/$O/bench.log 2>&1)
find $O/bench -name '*kernel_stats.csv' -exec cp {} $O/bench_kernel_stats.csv \;
rm -rf $O/bench
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/qkv -- python $R/tools/gemm_one.py 256 65792 4224 1408 200 > $R/$O/qkv.log 2>&1)
find $O/qkv -name '*kernel_stats.csv' -exec cp {} $O/qkv_gemm256_kernel_stats.csv \;
rm -rf $O/qkv
timeout 600 bash tools/pmc_qkv.sh > $O/pmc_qkv.log 2>&1
cp gpurun_out/pmc_qkv_summary.json $O/pmc_qkv_summary.json
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/decode -- python $R/tools/decode_only.py > $R/$O/decode.log 2>&1)
find $O/decode -name '*kernel_stats.csv' -exec cp {} $O/decode_kernel_stats.csv \;
rm -rf $O/decode
timeout 300 bash tools/pmc_skinny.sh > $O/pmc_skinny.log 2>&1
cp gpurun_out/pmc_skinny_summary.json $O/pmc_decode_gemm.json
timeout 300 bash tools/pmc_skinny.sh 22016 4096 swiglu > $O/pmc_skinny_gate_up.log 2>&1
cp gpurun_out/pmc_skinny_22016_summary.json $O/pmc_decode_gemm_gate_up.json
timeout 300 bash tools/pmc_attn.sh > $O/pmc_attn.log 2>&1
cp gpurun_out/pmc_attn_summary.json $O/pmc_attention.json 2>/dev/null
timeout 900 python bench.py > $O/bench_final.json 2> $O/bench_final.err
echo done
