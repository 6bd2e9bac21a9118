#!/bin/bash
# round 4, call 7: ViT attention, store-tolerant K / Q wait (attn_store_wait) A/B, isolated and in the tokenize pass
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
O=gpurun_out/r04c7
mkdir -p $O
timeout 300 python tools/attn_store_wait_ab.py > $O/attn_ab.log 2>&1; echo "attn rc=$?" >> $O/attn_ab.log
OUT=$O/tok_ab.json ROUNDS=5 timeout 600 python tools/tok_ab.py "attn_store_wait=1" "attn_store_wait=0" > $O/tok_ab.log 2>&1; echo "tok rc=$?" >> $O/tok_ab.log
grep -v "^/opt" $O/attn_ab.log | tail -6; python - <<'PY'
import json
d = json.load(open("gpurun_out/r04c7/tok_ab.json"))
for k, v in d.items():
    if isinstance(v, dict): print(k, v.get("median_ms"), v.get("img_s"), v.get("ids_equal_to_first"))
PY
