#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
mkdir -p gpurun_out/r02
for V in default late; do
  if [ $V = late ]; then export SEEDMI_LIB_PATH=$R/seed_amd/libseedmi_late.so; else unset SEEDMI_LIB_PATH; fi
  VARIANTS=256 REPS=40 timeout 300 python tools/gemm_sustained.py > gpurun_out/r02/gemm_sustained_run6_$V.log 2>&1
  timeout 400 python tools/tok_ab.py "tokenize_streams=2" "tokenize_streams=2,gemm_residual_nt=0" "tokenize_streams=1" > gpurun_out/r02/tok_ab6_$V.log 2>&1
done
unset SEEDMI_LIB_PATH
echo done
