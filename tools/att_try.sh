#!/bin/bash
# Is the SQ thread trace (rocprofv3 --att) usable on this image?  (VERDICT r5 item 2 asks for one; the decoder library
# librocprof-trace-decoder is not part of the ROCm 7.2.0 image.)  Writes gpurun_out/att_try.log.
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
find / -name "*trace-decoder*" -o -name "*trace_decoder*" 2>/dev/null | head
timeout 170 rocprofv3 --att --att-target-cu 1 --kernel-trace -d /tmp/att_out -- python $R/tools/gemm_one.py 256 65792 4224 1408 1 2>&1 | tail -25
find /tmp/att_out -type f | head -20
