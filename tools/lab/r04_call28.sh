#!/bin/bash
# round 4, call 28: GEMM schedule 12369 = 8273 + s_setprio 1 during a wave's LOAD sections (0 during its MFMA sections)
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
O=gpurun_out/r04c28
mkdir -p $O
export TMPDIR=/tmp
SCHEDS=8273,12369 SHAPES=qkv,proj,fc1,fc2 ROUNDS=8 OUT=$O/gemm_sched_ab.json timeout 600 python tools/gemm_sched_ab.py > $O/gemm_sched_ab.log 2>&1; echo "ab rc=$?" >> $O/gemm_sched_ab.log
OUT=$O/tok_ab.json ROUNDS=7 timeout 800 python tools/tok_ab.py "" "gemm_sched=12369" > $O/tok_ab.log 2>&1; echo "tok rc=$?" >> $O/tok_ab.log
grep -v "^/opt" $O/gemm_sched_ab.log | cut -c1-260 | tail -6
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04c28/tok_ab.json"))
for k, v in d.items():
    if isinstance(v, dict): print(repr(k), v.get("median_ms"), v.get("img_s"))
PY
