#!/bin/bash
# round 4, call 2: the new parity / boundary tests (after the sticky-word fix), the 32x32x16 two-phase K-tile A/B
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
O=gpurun_out/r04c2
mkdir -p $O
export TMPDIR=/tmp
SEEDMI_LIB_PATH=seed_amd/libseedmi_dev.so OUT=$O/gemm_mfma32_ab.json timeout 600 python tools/gemm_mfma32_ab.py > $O/gemm_mfma32_ab.log 2>&1; echo "ab rc=$?" >> $O/gemm_mfma32_ab.log
timeout 1500 python -m pytest tests -m gpu -q -s -k "14b or peaked or live_reference or fork_join or sticky or padded or end_to_end_on_device or full_depth_config3 or batching or serve" > $O/pytest_new.log 2>&1; echo "pytest rc=$?" >> $O/pytest_new.log
cat $O/gemm_mfma32_ab.log | tail -8
grep -E "^\[|passed|failed|rc=" $O/pytest_new.log | tail -40
