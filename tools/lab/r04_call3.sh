#!/bin/bash
# round 4, call 3: two-phase K-tile with the ragged last n-tile re-divided (schedule 337, devtools) vs the default (81): isolated GEMMs and
# the tokenize pass; the two tests that failed in call 2
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
O=gpurun_out/r04c3
mkdir -p $O
export TMPDIR=/tmp
SEEDMI_LIB_PATH=seed_amd/libseedmi_dev.so SCHEDS=81,337 SHAPES=qkv,proj,fc1,fc2 ROUNDS=5 OUT=$O/gemm_sched_ab.json timeout 600 python tools/gemm_sched_ab.py > $O/gemm_sched_ab.log 2>&1; echo "ab rc=$?" >> $O/gemm_sched_ab.log
SEEDMI_LIB_PATH=seed_amd/libseedmi_dev.so OUT=$O/tok_ab.json ROUNDS=5 timeout 600 python tools/tok_ab.py "" "gemm_sched=337" > $O/tok_ab.log 2>&1; echo "tok rc=$?" >> $O/tok_ab.log
timeout 900 python -m pytest tests -m gpu -q -s -k "14b or fork_join" > $O/pytest_two.log 2>&1; echo "pytest rc=$?" >> $O/pytest_two.log
tail -6 $O/gemm_sched_ab.log; tail -8 $O/tok_ab.log
grep -E "^\[|passed|failed|rc=" $O/pytest_two.log | tail -15
