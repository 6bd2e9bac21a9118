#!/bin/bash
# round 4, call 24: staggered attention - last key tile's three dead registers skipped in the softmax; 8 = side scores in the QK^T slot (two side rows)
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
O=gpurun_out/r04c24
mkdir -p $O
export TMPDIR=/tmp
MODES=3,5,8,4,6 BATCHES=72,128 ROUNDS=8 timeout 300 python tools/attn_modes_ab.py > $O/attn_modes_ab.log 2>&1; echo "rc=$?" >> $O/attn_modes_ab.log
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "attention_fullrow" > $O/pytest_attn.log 2>&1; echo "pytest rc=$?" >> $O/pytest_attn.log
grep -v "^/opt" $O/attn_modes_ab.log | tail -12
tail -2 $O/pytest_attn.log
