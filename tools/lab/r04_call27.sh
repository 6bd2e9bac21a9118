#!/bin/bash
# round 4, call 27: staggered attention, more priority orders {QK, SM, PV}: 5 = {1,3,2} (shipped), 8 = {1,3,3}, 9 = {1,2,2}, 10 = {0,3,1}, 11 = {2,3,3}, 12 = {1,3,1}
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
O=gpurun_out/r04c27
mkdir -p $O
export TMPDIR=/tmp
MODES=5,8,9,10,11,12 BATCHES=128 ROUNDS=8 timeout 300 python tools/attn_modes_ab.py > $O/attn_modes_ab.log 2>&1; echo "rc=$?" >> $O/attn_modes_ab.log
grep -v "^/opt" $O/attn_modes_ab.log | tail -8
