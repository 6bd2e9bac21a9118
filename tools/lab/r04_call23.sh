#!/bin/bash
# round 4, call 23: staggered attention, final priorities: 5 = {1,3,2} with group B's QK at 2, 8 = group B's QK at 1 too, 7 = none; end to end
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
O=gpurun_out/r04c23
mkdir -p $O
export TMPDIR=/tmp
MODES=3,5,7,8,6 BATCHES=128 ROUNDS=8 timeout 300 python tools/attn_modes_ab.py > $O/attn_modes_ab.log 2>&1; echo "rc=$?" >> $O/attn_modes_ab.log
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "attention_fullrow" > $O/pytest_attn.log 2>&1; echo "pytest rc=$?" >> $O/pytest_attn.log
OUT=$O/tok_ab.json ROUNDS=9 timeout 800 python tools/tok_ab.py "" "attn_vit=8" "attn_vit=3" > $O/tok_ab.log 2>&1; echo "tok rc=$?" >> $O/tok_ab.log
grep -v "^/opt" $O/attn_modes_ab.log | tail -8
tail -2 $O/pytest_attn.log
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04c23/tok_ab.json"))
for k, v in d.items():
    if isinstance(v, dict): print(repr(k), v.get("median_ms"), v.get("img_s"))
PY
