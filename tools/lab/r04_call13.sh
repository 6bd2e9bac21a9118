#!/bin/bash
# round 4, call 13: schedule 8273 as the default - LLaMA prefill epilogues (8257 vs 65) A/B, GEMM + tokenizer tests under the new default
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
O=gpurun_out/r04c13
mkdir -p $O
export TMPDIR=/tmp
AB="gemm_sched=81|gemm_sched=-1" ROUNDS=7 timeout 600 python tools/prefill_bench.py > $O/prefill_ab_14b.log 2>&1; echo "rc=$?" >> $O/prefill_ab_14b.log
MODEL=8b AB="gemm_sched=81|gemm_sched=-1" ROUNDS=7 timeout 600 python tools/prefill_bench.py > $O/prefill_ab_8b.log 2>&1; echo "rc=$?" >> $O/prefill_ab_8b.log
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_tokenizer.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
grep -v "^/opt" $O/prefill_ab_14b.log | tail -4; grep -v "^/opt" $O/prefill_ab_8b.log | tail -4; tail -4 $O/pytest.log
