#!/bin/bash
# round 4, call 20: staggered attention, wave priority schemes per phase {QK, SM, PV}: 7 = 3/2/1, 8 = 2/3/1, 9 = 3/1/2, 10 = 1/3/2, 11 = 1/2/3
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
O=gpurun_out/r04c20
mkdir -p $O
export TMPDIR=/tmp
MODES=3,5,7,8,9,10,11 BATCHES=128 ROUNDS=6 timeout 300 python tools/attn_modes_ab.py > $O/attn_modes_ab.log 2>&1; echo "rc=$?" >> $O/attn_modes_ab.log
grep -v "^/opt" $O/attn_modes_ab.log | tail -12
