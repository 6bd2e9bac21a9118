#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd $R
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
{
echo "=== K-tile phase stamps: schedule 31 (buffer requests, static slots) vs 543 (flat requests) vs 0"; date
for sh in qkv fc2; do SEEDMI_LIB_PATH=seed_amd/libseedmi_dev.so SCHEDS=31,543,0 SHAPE=$sh timeout 300 python tools/gemm_phase_times.py 2>&1 | tail -40; done
date
} > gpurun_out/r03/call24.log 2>&1
tail -90 gpurun_out/r03/call24.log
