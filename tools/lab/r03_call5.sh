#!/bin/bash
# round 3, GPU call 5: 16-wave attention modes (wide stores, flash normalisation), streams check
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
{
echo "=== attention tests"; date
timeout 600 python -m pytest -q -m gpu -s "tests/test_gpu_kernels.py::test_attention_fullrow" 2>&1 | grep -E "trv=[345]|side path|passed|failed|Error|assert" | tail -20
echo "=== attention bench"; date
B=128 timeout 300 python tools/attn_bench.py 2>&1 | tail -6
B=256 timeout 300 python tools/attn_bench.py 2>&1 | tail -6
echo "=== end-to-end A/B"; date
ROUNDS=4 OUT=gpurun_out/r03/tok_ab_call5.json timeout 600 python tools/tok_ab.py "" "attn_vit=3" "attn_vit=4" "tokenize_streams=1" "tokenize_streams=1,attn_vit=4" 2>&1 | python -c "
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
} > gpurun_out/r03/call5.log 2>&1
tail -70 gpurun_out/r03/call5.log
