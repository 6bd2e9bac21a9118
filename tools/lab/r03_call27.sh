#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd $R
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
{
echo "=== two-phase K-tile: fragment waits in front of (81) / behind (2129) the barriers"; date
SEEDMI_LIB_PATH=seed_amd/libseedmi_dev.so SCHEDS=0,81,2129,31 ROUNDS=6 OUT=gpurun_out/r03/gemm_sched_ab_call27.json timeout 600 python tools/gemm_sched_ab.py 2>&1 | tail -5
SEEDMI_LIB_PATH=seed_amd/libseedmi_dev.so ROUNDS=4 OUT=gpurun_out/r03/tok_ab_call27.json timeout 600 python tools/tok_ab.py "" "gemm_sched=2129" 2>&1 | python -c "
import sys, json
t = sys.stdin.read()
try:
    d = json.loads(t[t.index('{\n'):])
    for k, v in d.items(): print(repr(k), v['median_ms'], v['img_s'], v['all_ms'])
except Exception as e:
    print(t[-3000:])
"
date
} > gpurun_out/r03/call27.log 2>&1
tail -30 gpurun_out/r03/call27.log
