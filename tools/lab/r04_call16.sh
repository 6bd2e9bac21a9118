#!/bin/bash
# round 4, call 16: staggered attention end to end (more rounds), with / without wave priorities
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
O=gpurun_out/r04c16
mkdir -p $O
export TMPDIR=/tmp
OUT=$O/tok_ab.json ROUNDS=11 timeout 800 python tools/tok_ab.py "" "attn_vit=5" "attn_vit=7" > $O/tok_ab.log 2>&1; echo "tok rc=$?" >> $O/tok_ab.log
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04c16/tok_ab.json"))
for k, v in d.items():
    if isinstance(v, dict): print(repr(k), v.get("median_ms"), v.get("img_s"), v.get("all_ms"))
PY
tail -3 $O/tok_ab.log
