#!/bin/bash
# round 3, GPU call 4: full GPU suite under the new default schedule, tile statistics end to end
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
{
echo "=== full GPU suite"; date
timeout 1200 python -m pytest -q -m gpu tests 2>&1 | tail -25
echo "=== end-to-end A/B: tile statistics"; date
ROUNDS=4 OUT=gpurun_out/r03/tok_ab_call4.json timeout 600 python tools/tok_ab.py "gemm_sched=0" "" "tokenize_tile_stats=1" "tokenize_tile_stats=1,attn_vit=2" 2>&1 | python -c "
import sys, json
t = sys.stdin.read()
try:
    d = json.loads(t[t.index('{\n'):])
    print(t[:t.index('{\n')][-1500:])
    for k, v in d.items(): print(repr(k), v['median_ms'], v['img_s'], v['all_ms'])
except Exception as e:
    print(t[-3000:])
"
date
} > gpurun_out/r03/call4.log 2>&1
tail -70 gpurun_out/r03/call4.log
