#!/bin/bash
# round 3, GPU call 1: new parity tests, GEMM schedule variants (bit-identity + interleaved timing), phase clock stamps, end-to-end A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
{
echo "=== new / changed GPU tests"; date
timeout 900 python -m pytest -q -m gpu -s \
  "tests/test_gpu_llama.py::test_llama8b_full_depth_config3_parity" \
  "tests/test_gpu_surfaces.py::test_reference_fixture_cat_jpg_both_preprocessing_routes" \
  "tests/test_gpu_tokenizer.py::test_rccl_id_gather_on_every_visible_gpu" \
  tests/test_batching.py \
  "tests/test_gpu_kernels.py::test_gemm_streamk_is_bit_identical_to_data_parallel" \
  "tests/test_gpu_kernels.py::test_vq_argmin_bit_exact" 2>&1 | tail -40
echo "=== gemm schedule A/B"; date
timeout 600 python tools/gemm_sched_ab.py 2>&1 | tail -20
echo "=== phase stamps (devtools build)"; date
for sh in qkv proj; do SEEDMI_LIB_PATH=seed_amd/libseedmi_dev.so SCHEDS=0,1,3,7 SHAPE=$sh timeout 300 python tools/gemm_phase_times.py 2>&1 | tail -30; done
echo "=== end-to-end A/B"; date
ROUNDS=4 timeout 600 python tools/tok_ab.py "gemm_sched=0" "gemm_sched=1" "gemm_sched=3" "gemm_sched=7" 2>&1 | tail -12
date
} > gpurun_out/r03_call1.log 2>&1
tail -120 gpurun_out/r03_call1.log
