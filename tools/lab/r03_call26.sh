#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd $R
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
{
echo "=== GEMM + tokenizer tests under the two-phase default (81)"; date
timeout 900 python -m pytest -q -x -m gpu tests/test_gpu_kernels.py -k "gemm or layernorm or statistics" 2>&1 | tail -3
timeout 900 python -m pytest -q -x -m gpu tests/test_gpu_tokenizer.py tests/test_detokenizer.py 2>&1 | tail -3
echo "=== schedules: 81 default, 593 = 81 with flat requests, 113 = 81 + static priority, 65 = 81 without early residual rows, 31, 0"; date
SEEDMI_LIB_PATH=seed_amd/libseedmi_dev.so SCHEDS=0,31,81,593,113,65 ROUNDS=5 OUT=gpurun_out/r03/gemm_sched_ab_call26.json timeout 600 python tools/gemm_sched_ab.py 2>&1 | tail -5
SEEDMI_LIB_PATH=seed_amd/libseedmi_dev.so ROUNDS=4 OUT=gpurun_out/r03/tok_ab_call26.json timeout 600 python tools/tok_ab.py "" "gemm_sched=31" "gemm_sched=593" "gemm_sched=113" "gemm_sched=0" "attn_vit=4" "tokenize_streams=1" 2>&1 | python -c "
import sys, json
t = sys.stdin.read()
try:
    d = json.loads(t[t.index('{\n'):])
    for k, v in d.items(): print(repr(k), v['median_ms'], v['img_s'], v['all_ms'])
except Exception as e:
    print(t[-3000:])
"
echo "=== phase stamps of the two-phase K-tile"; date
SEEDMI_LIB_PATH=seed_amd/libseedmi_dev.so SCHEDS=81 SHAPE=qkv timeout 300 python tools/gemm_phase_times.py 2>&1 | tail -8
date
} > gpurun_out/r03/call26.log 2>&1
tail -40 gpurun_out/r03/call26.log
