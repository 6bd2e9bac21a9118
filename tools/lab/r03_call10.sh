#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
{
echo "=== decode GEMM ablations (devtools library)"; date
SEEDMI_LIB_PATH=seed_amd/libseedmi_dev.so timeout 300 python tools/skinny_ablate.py 2>&1 | tail -8
date
} > gpurun_out/r03/call10.log 2>&1
tail -60 gpurun_out/r03/call10.log
