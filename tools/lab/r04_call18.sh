#!/bin/bash
# round 4, call 18: staggered attention, XCD-aware item walk (all heads of an image on one XCD)
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
O=gpurun_out/r04c18
mkdir -p $O
export TMPDIR=/tmp
MODES=3,5,5:xcd0,6 BATCHES=40,100,128 ROUNDS=6 timeout 300 python tools/attn_modes_ab.py > $O/attn_modes_ab.log 2>&1; echo "rc=$?" >> $O/attn_modes_ab.log
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "attention_fullrow" > $O/pytest_attn.log 2>&1; echo "pytest rc=$?" >> $O/pytest_attn.log
OUT=$O/tok_ab.json ROUNDS=9 timeout 800 python tools/tok_ab.py "" "attn_xcd=0" "attn_vit=3" > $O/tok_ab.log 2>&1; echo "tok rc=$?" >> $O/tok_ab.log
grep -v "^/opt" $O/attn_modes_ab.log | tail -12
tail -3 $O/pytest_attn.log
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04c18/tok_ab.json"))
for k, v in d.items():
    if isinstance(v, dict): print(repr(k), v.get("median_ms"), v.get("img_s"))
PY
