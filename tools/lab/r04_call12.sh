#!/bin/bash
# round 4, call 12: position-free body (8273) and its pair-unrolled static-slot form (24657) vs 81, more rounds
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
O=gpurun_out/r04c12
mkdir -p $O
export TMPDIR=/tmp
SCHEDS=81,8273,24657 SHAPES=qkv,proj,fc1,fc2 ROUNDS=8 OUT=$O/gemm_sched_ab.json timeout 600 python tools/gemm_sched_ab.py > $O/gemm_sched_ab.log 2>&1; echo "ab rc=$?" >> $O/gemm_sched_ab.log
OUT=$O/tok_ab.json ROUNDS=7 timeout 800 python tools/tok_ab.py "" "gemm_sched=8273" "gemm_sched=24657" > $O/tok_ab.log 2>&1; echo "tok rc=$?" >> $O/tok_ab.log
grep -v "^/opt" $O/gemm_sched_ab.log | cut -c1-330 | tail -8
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04c12/tok_ab.json"))
for k, v in d.items():
    if isinstance(v, dict): print(repr(k), v.get("median_ms"), v.get("img_s"), {kk: vv for kk, vv in v.items() if "equal" in kk or "ident" in kk})
PY
