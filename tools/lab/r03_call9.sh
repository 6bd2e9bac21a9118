#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
{
echo "=== attention tests"; date
timeout 600 python -m pytest -q -m gpu -s "tests/test_gpu_kernels.py::test_attention_fullrow" 2>&1 | grep -E "trv=[67]|side path|passed|failed|Error|assert" | tail -20
echo "=== attention bench"; date
B=128 timeout 300 python tools/attn_bench.py 2>&1 | tail -7
B=256 timeout 300 python tools/attn_bench.py 2>&1 | tail -7
echo "=== end-to-end A/B"; date
ROUNDS=4 OUT=gpurun_out/r03/tok_ab_call9.json timeout 600 python tools/tok_ab.py "" "attn_vit=5" "attn_vit=6" 2>&1 | python -c "
import sys, json
t = sys.stdin.read()
try:
    d = json.loads(t[t.index('{\n'):])
    for k, v in d.items(): print(repr(k), v['median_ms'], v['img_s'], v['all_ms'])
except Exception as e:
    print(t[-3000:])
"
date
} > gpurun_out/r03/call9.log 2>&1
tail -60 gpurun_out/r03/call9.log
