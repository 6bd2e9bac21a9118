#!/bin/bash
# round 4, call 32: product 256x256 kernel with W read as if packed request-major at load time (schedule 12369, devtools, TIMING ONLY)
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
O=gpurun_out/r04c32
mkdir -p $O
export TMPDIR=/tmp
SEEDMI_LIB_PATH=seed_amd/libseedmi_dev.so SCHEDS=8273,12369 SHAPES=qkv,proj,fc1,fc2 ROUNDS=8 OUT=$O/gemm_sched_ab.json timeout 600 python tools/gemm_sched_ab.py > $O/gemm_sched_ab.log 2>&1; echo "ab rc=$?" >> $O/gemm_sched_ab.log
grep -v "^/opt" $O/gemm_sched_ab.log | cut -c1-260 | tail -6
