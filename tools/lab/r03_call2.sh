#!/bin/bash
# round 3, GPU call 2: 16-wave ViT attention, balanced GEMM schedules, reworked phase stamps, end-to-end A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
{
echo "=== tests"; date
timeout 900 python -m pytest -q -m gpu -s \
  "tests/test_gpu_llama.py::test_llama8b_full_depth_config3_parity" \
  "tests/test_gpu_kernels.py::test_attention_fullrow" 2>&1 | grep -v "^\[attention hd64\|^\[attention hd88 1" | tail -40
echo "=== attention bench"; date
B=128 timeout 300 python tools/attn_bench.py 2>&1 | tail -5
B=256 timeout 300 python tools/attn_bench.py 2>&1 | tail -5
echo "=== gemm schedule A/B"; date
SCHEDS=0,7,13,15,31,63 OUT=gpurun_out/r03/gemm_sched_ab_call2.json timeout 600 python tools/gemm_sched_ab.py 2>&1 | tail -8
echo "=== phase stamps (devtools build)"; date
SEEDMI_LIB_PATH=seed_amd/libseedmi_dev.so SCHEDS=0,15 SHAPE=qkv timeout 300 python tools/gemm_phase_times.py 2>&1 | tail -12
SEEDMI_LIB_PATH=seed_amd/libseedmi_dev.so SCHEDS=0,31 SHAPE=proj timeout 300 python tools/gemm_phase_times.py 2>&1 | tail -12
SEEDMI_LIB_PATH=seed_amd/libseedmi_dev.so SCHEDS=15 SHAPE=fc1 timeout 300 python tools/gemm_phase_times.py 2>&1 | tail -6
echo "=== end-to-end A/B"; date
ROUNDS=4 OUT=gpurun_out/r03/tok_ab_call2.json timeout 600 python tools/tok_ab.py "gemm_sched=0" "gemm_sched=15" "gemm_sched=31" "gemm_sched=31,attn_vit=2" "gemm_sched=63,attn_vit=2" 2>&1 | python -c "
import sys, json, re
t = sys.stdin.read()
try:
    d = json.loads(t[t.index('{'):])
    for k, v in d.items(): print(k, v['median_ms'], v['img_s'], v['all_ms'])
except Exception as e:
    print(t[-3000:])
"
date
} > gpurun_out/r03/call2.log 2>&1
tail -150 gpurun_out/r03/call2.log
