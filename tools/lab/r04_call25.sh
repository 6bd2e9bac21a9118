#!/bin/bash
# round 4, call 25: staggered attention vs lock-step kernel, bit-identity over batch sizes around the XCD-walk threshold and uneven images per XCD
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
O=gpurun_out/r04c25
mkdir -p $O
export TMPDIR=/tmp
MODES=3,5,5:xcd0 BATCHES=16,63,64,65,71,127,129,200,256 ROUNDS=2 REPS=6 timeout 600 python tools/attn_modes_ab.py > $O/attn_modes_ab.log 2>&1; echo "rc=$?" >> $O/attn_modes_ab.log
grep -v "^/opt" $O/attn_modes_ab.log | tail -32
