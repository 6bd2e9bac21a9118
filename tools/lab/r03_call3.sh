#!/bin/bash
# round 3, GPU call 3: two-phase GEMM schedule - parity suite under it, race-screened A/B, end to end
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
{
for v in 81 31; do
echo "=== GEMM parity tests under gemm_sched=$v"; date
SEEDMI_OPTIONS="gemm_sched=$v" timeout 900 python -m pytest -q -m gpu tests/test_gpu_kernels.py -k "gemm or patch_embed" 2>&1 | tail -6
done
echo "=== gemm schedule A/B"; date
SCHEDS=0,31,81,113 OUT=gpurun_out/r03/gemm_sched_ab_call3.json timeout 600 python tools/gemm_sched_ab.py 2>&1 | tail -8
echo "=== end-to-end A/B"; date
ROUNDS=4 OUT=gpurun_out/r03/tok_ab_call3.json timeout 600 python tools/tok_ab.py "gemm_sched=0" "gemm_sched=31" "gemm_sched=81" "gemm_sched=113" 2>&1 | python -c "
import sys, json
t = sys.stdin.read()
try:
    d = json.loads(t[t.index('{'):])
    for k, v in d.items(): print(k, v['median_ms'], v['img_s'], v['all_ms'])
except Exception as e:
    print(t[-3000:])
"
date
} > gpurun_out/r03/call3.log 2>&1
tail -60 gpurun_out/r03/call3.log
