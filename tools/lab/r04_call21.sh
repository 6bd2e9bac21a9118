#!/bin/bash
# round 4, call 21: staggered attention, side-row PV dealt over six waves; priority scheme 10; stamps
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
O=gpurun_out/r04c21
mkdir -p $O
export TMPDIR=/tmp
MODES=3,5,10,8,6 BATCHES=3,128 ROUNDS=6 timeout 300 python tools/attn_modes_ab.py > $O/attn_modes_ab.log 2>&1; echo "rc=$?" >> $O/attn_modes_ab.log
SEEDMI_LIB_PATH=seed_amd/libseedmi_dev.so MODES=5,10 timeout 300 python tools/attn16s_phase_times.py > $O/phase_times.log 2>&1; echo "rc=$?" >> $O/phase_times.log
grep -v "^/opt" $O/attn_modes_ab.log | tail -12
grep -v "^/opt" $O/phase_times.log | cut -c1-200 | grep -v "wave 1:\|wave  7\|wave  9\|wave 15"
