#!/bin/bash
# round 4, call 26: XCD-aware walk below 64 images (attn_xcd = 2: from 16 images)
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
O=gpurun_out/r04c26
mkdir -p $O
export TMPDIR=/tmp
MODES=5,5:xcd2 BATCHES=16,24,32,40,48,56,63 ROUNDS=5 REPS=20 timeout 600 python tools/attn_modes_ab.py > $O/attn_modes_ab.log 2>&1; echo "rc=$?" >> $O/attn_modes_ab.log
grep -v "^/opt" $O/attn_modes_ab.log | tail -32
