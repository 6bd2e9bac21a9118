#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
{
echo "=== 16-wave attention phase stamps"; date
SEEDMI_LIB_PATH=seed_amd/libseedmi_dev.so MODES=3,4 timeout 300 python tools/attn16_phase_times.py 2>&1 | tail -20
date
} > gpurun_out/r03/call8.log 2>&1
tail -40 gpurun_out/r03/call8.log
