#!/bin/bash
# round 3, GPU call 7: ragged last n-tile re-divided (one K-tile body, uniform branches) - parity, A/B, end to end; kernel timelines
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
{
echo "=== GEMM parity tests under gemm_sched=287"; date
SEEDMI_OPTIONS="gemm_sched=287" timeout 900 python -m pytest -q -m gpu tests/test_gpu_kernels.py -k "gemm or patch_embed or layernorm_statistics" 2>&1 | tail -6
echo "=== gemm schedule A/B"; date
SCHEDS=0,31,287 OUT=gpurun_out/r03/gemm_sched_ab_call7.json timeout 600 python tools/gemm_sched_ab.py 2>&1 | tail -8
echo "=== end-to-end A/B"; date
ROUNDS=4 OUT=gpurun_out/r03/tok_ab_call7.json timeout 600 python tools/tok_ab.py "" "gemm_sched=287" "gemm_sched=287,attn_vit=4" 2>&1 | python -c "
import sys, json
t = sys.stdin.read()
try:
    d = json.loads(t[t.index('{\n'):])
    for k, v in d.items(): print(repr(k), v['median_ms'], v['img_s'], v['all_ms'])
except Exception as e:
    print(t[-3000:])
"
for st in 2 1; do
echo "=== kernel timeline, tokenize_streams=$st"; date
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt$st -- python $R/tools/tok_trace.py tokenize_streams=$st > /tmp/kt$st.log 2>&1)
f=$(find /tmp/kt$st -name '*kernel_trace.csv' | head -1)
python tools/kernel_timeline.py $f gpurun_out/r03/kt_compact_$st.csv && NPARTS=$st python tools/timeline_stats.py gpurun_out/r03/kt_compact_$st.csv | tee gpurun_out/r03/timeline_streams_$st.txt
rm -f gpurun_out/r03/kt_compact_$st.csv
done
date
} > gpurun_out/r03/call7.log 2>&1
tail -90 gpurun_out/r03/call7.log
