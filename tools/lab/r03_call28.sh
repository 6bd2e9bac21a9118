#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd $R
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
{
echo "=== tests: two-phase K-tile for the plain / SwiGLU / tanh / ReLU epilogues too"; date
timeout 1200 python -m pytest -q -x -m gpu tests/test_gpu_kernels.py tests/test_gpu_llama.py tests/test_detokenizer.py -k "not full_depth" 2>&1 | tail -3
echo "=== 14B prefill (config 5 share of one GPU): default vs gemm_sched=31 (those epilogues on schedule 0)"; date
timeout 600 python tools/prefill_bench.py 2>&1 | tail -3
GEMM_SCHED=31 timeout 600 python - <<'PY' 2>&1 | tail -3
import os, sys, runpy
sys.path.insert(0, os.getcwd())
from seed_amd import lib as L
L.check(L.load().seedmi_set_option(b"gemm_sched", 31), "opt")
sys.argv = ["tools/prefill_bench.py"]
runpy.run_path("tools/prefill_bench.py", run_name="__main__")
PY
date
} > gpurun_out/r03/call28.log 2>&1
tail -30 gpurun_out/r03/call28.log
