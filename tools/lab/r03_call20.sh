#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd $R
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
{
echo "=== GEMM parity tests (ring slots 32 KiB apart, compile-time slots in the QKV / fc1 variants, even stream-K cuts)"; date
timeout 900 python -m pytest -q -x -m gpu tests/test_gpu_kernels.py -k "gemm or layernorm or statistics" 2>&1 | tail -4
echo "=== gemm schedule A/B: 31 = default, 1055 = 31 with run-time ring slots, 543 = 31 with flat requests"; date
SEEDMI_LIB_PATH=seed_amd/libseedmi_dev.so SCHEDS=0,31,1055,543 ROUNDS=7 OUT=gpurun_out/r03/gemm_sched_ab_call20.json timeout 600 python tools/gemm_sched_ab.py 2>&1 | tail -5
echo "=== end-to-end A/B"; date
SEEDMI_LIB_PATH=seed_amd/libseedmi_dev.so ROUNDS=5 OUT=gpurun_out/r03/tok_ab_call20.json timeout 600 python tools/tok_ab.py "" "gemm_sched=1055" "gemm_sched=543" 2>&1 | python -c "
import sys, json
t = sys.stdin.read()
try:
    d = json.loads(t[t.index('{\n'):])
    for k, v in d.items(): print(repr(k), v['median_ms'], v['img_s'], v['all_ms'])
except Exception as e:
    print(t[-3000:])
"
echo "=== tokenizer tests"; date
timeout 900 python -m pytest -q -x -m gpu tests/test_gpu_tokenizer.py 2>&1 | tail -3
date
} > gpurun_out/r03/call20.log 2>&1
tail -40 gpurun_out/r03/call20.log
