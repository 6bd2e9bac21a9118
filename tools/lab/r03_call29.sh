#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd $R
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
{
echo "=== clocks and package power while the tokenize pass runs (rocm-smi samples, 1 s apart)"; date
rocm-smi --showpower --showclocks 2>&1 | grep -E "Power|sclk|mclk|fclk" | head -8
(ROUNDS=14 OUT=gpurun_out/r03/tok_ab_call29.json timeout 300 python tools/tok_ab.py "" "gemm_sched=31" "gemm_sched=0" > gpurun_out/r03/tok_ab_call29.log 2>&1) &
BG=$!
sleep 45
for i in 1 2 3 4 5 6 7 8 9 10 11 12; do rocm-smi --showpower --showclocks 2>&1 | grep -E "Power|sclk" | tr '\n' ' '; echo; sleep 1; done
wait $BG
python - <<'PY'
import json
d = json.load(open("gpurun_out/r03/tok_ab_call29.json"))
for k, v in d.items(): print(repr(k), v["median_ms"], v["img_s"], v["all_ms"])
PY
date
} > gpurun_out/r03/call29.log 2>&1
tail -40 gpurun_out/r03/call29.log
