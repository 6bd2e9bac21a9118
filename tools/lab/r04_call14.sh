#!/bin/bash
# round 4, call 14: staggered 16-wave ViT attention (attn_vit = 5 / 6) vs the lock-step kernel (3 / 4)
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
O=gpurun_out/r04c14
mkdir -p $O
export TMPDIR=/tmp
BATCHES=3,128 timeout 300 python tools/attn_modes_ab.py > $O/attn_modes_ab.log 2>&1; echo "rc=$?" >> $O/attn_modes_ab.log
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "attention_fullrow" > $O/pytest_attn.log 2>&1; echo "pytest rc=$?" >> $O/pytest_attn.log
OUT=$O/tok_ab.json ROUNDS=5 timeout 600 python tools/tok_ab.py "" "attn_vit=5" "attn_vit=4" "attn_vit=6" > $O/tok_ab.log 2>&1; echo "tok rc=$?" >> $O/tok_ab.log
grep -v "^/opt" $O/attn_modes_ab.log | tail -12
tail -3 $O/pytest_attn.log
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04c14/tok_ab.json"))
for k, v in d.items():
    if isinstance(v, dict): print(repr(k), v.get("median_ms"), v.get("img_s"), {kk: vv for kk, vv in v.items() if "equal" in kk or "ident" in kk or "group" in kk})
PY
