#!/bin/bash
# round 4, call 11: position-free K-tile body (schedule 8273 = 81 + bit 13) vs the default 81
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
O=gpurun_out/r04c11
mkdir -p $O
export TMPDIR=/tmp
SCHEDS=81,8273 SHAPES=qkv,proj,fc1,fc2 ROUNDS=6 OUT=$O/gemm_sched_ab.json timeout 600 python tools/gemm_sched_ab.py > $O/gemm_sched_ab.log 2>&1; echo "ab rc=$?" >> $O/gemm_sched_ab.log
OUT=$O/tok_ab.json ROUNDS=5 timeout 600 python tools/tok_ab.py "" "gemm_sched=8273" > $O/tok_ab.log 2>&1; echo "tok rc=$?" >> $O/tok_ab.log
SEEDMI_OPTIONS=gemm_sched=8273 timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "gemm" > $O/pytest_gemm_8273.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gemm_8273.log
grep -v "^/opt" $O/gemm_sched_ab.log | cut -c1-330 | tail -8
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04c11/tok_ab.json"))
for k, v in d.items():
    if isinstance(v, dict): print(repr(k), v.get("median_ms"), v.get("img_s"), {kk: vv for kk, vv in v.items() if "equal" in kk or "ident" in kk})
PY
tail -3 $O/pytest_gemm_8273.log
