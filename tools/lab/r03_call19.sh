#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd $R
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
{
echo "=== GEMM parity tests (buffer-form LDS-DMA requests)"; date
timeout 900 python -m pytest -q -x -m gpu tests/test_gpu_kernels.py -k "gemm or layernorm or statistics" 2>&1 | tail -4
echo "=== gemm schedule A/B: 0 / 31 with buffer requests, 512 / 543 = the same schedules with the flat requests of rounds 1-2"; date
SEEDMI_LIB_PATH=seed_amd/libseedmi_dev.so SCHEDS=0,31,512,543 OUT=gpurun_out/r03/gemm_sched_ab_call19.json timeout 600 python tools/gemm_sched_ab.py 2>&1 | tail -5
echo "=== end-to-end A/B"; date
SEEDMI_LIB_PATH=seed_amd/libseedmi_dev.so ROUNDS=4 OUT=gpurun_out/r03/tok_ab_call19.json timeout 600 python tools/tok_ab.py "" "gemm_sched=543" "gemm_sched=0" "gemm_sched=512" 2>&1 | python -c "
import sys, json
t = sys.stdin.read()
try:
    d = json.loads(t[t.index('{\n'):])
    for k, v in d.items(): print(repr(k), v['median_ms'], v['img_s'], v['all_ms'])
except Exception as e:
    print(t[-3000:])
"
date
} > gpurun_out/r03/call19.log 2>&1
tail -40 gpurun_out/r03/call19.log
