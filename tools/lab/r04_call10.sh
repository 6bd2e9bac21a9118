#!/bin/bash
# round 4, call 10: seam A/B with the control arm (seam code compiled, never taken)
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
O=gpurun_out/r04c10
mkdir -p $O
SCHEDS=81,8273,24657 SHAPES=qkv,fc2 ROUNDS=6 OUT=$O/gemm_sched_ab.json timeout 600 python tools/gemm_sched_ab.py > $O/gemm_sched_ab.log 2>&1; echo "ab rc=$?" >> $O/gemm_sched_ab.log
grep -v "^/opt" $O/gemm_sched_ab.log | cut -c1-500 | tail -4
