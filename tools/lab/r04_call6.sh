#!/bin/bash
# round 4, call 6: store-tolerant counted waits in the two-phase K-tile (schedule 4177 = 81 + bit 12) vs the default 81
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
O=gpurun_out/r04c6
mkdir -p $O
export TMPDIR=/tmp
SCHEDS=81,4177 SHAPES=qkv,proj,fc1,fc2 ROUNDS=6 OUT=$O/gemm_sched_ab.json timeout 600 python tools/gemm_sched_ab.py > $O/gemm_sched_ab.log 2>&1; echo "ab rc=$?" >> $O/gemm_sched_ab.log
OUT=$O/tok_ab.json ROUNDS=5 timeout 600 python tools/tok_ab.py "" "gemm_sched=4177" > $O/tok_ab.log 2>&1; echo "tok rc=$?" >> $O/tok_ab.log
grep -v "^/opt" $O/gemm_sched_ab.log | cut -c1-400 | tail -8; tail -12 $O/tok_ab.log
