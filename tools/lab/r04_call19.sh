#!/bin/bash
# round 4, call 19: slot stamps of the staggered attention with the XCD-aware walk
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
O=gpurun_out/r04c19
mkdir -p $O
export TMPDIR=/tmp
SEEDMI_LIB_PATH=seed_amd/libseedmi_dev.so MODES=5,6 timeout 300 python tools/attn16s_phase_times.py > $O/phase_times.log 2>&1; echo "rc=$?" >> $O/phase_times.log
grep -v "^/opt" $O/phase_times.log | cut -c1-230
