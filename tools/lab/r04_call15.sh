#!/bin/bash
# round 4, call 15: staggered attention - slot stamps, priority / deeper K prefetch variants
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
O=gpurun_out/r04c15
mkdir -p $O
export TMPDIR=/tmp
SEEDMI_LIB_PATH=seed_amd/libseedmi_dev.so MODES=5,6 timeout 300 python tools/attn16s_phase_times.py > $O/phase_times.log 2>&1; echo "rc=$?" >> $O/phase_times.log
SEEDMI_LIB_PATH=seed_amd/libseedmi_dev.so MODES=3 timeout 300 python tools/attn16_phase_times.py >> $O/phase_times.log 2>&1
MODES=3,5,7,8,9 BATCHES=128 ROUNDS=6 timeout 300 python tools/attn_modes_ab.py > $O/attn_modes_ab.log 2>&1; echo "rc=$?" >> $O/attn_modes_ab.log
grep -v "^/opt" $O/phase_times.log | cut -c1-230
grep -v "^/opt" $O/attn_modes_ab.log | tail -8
