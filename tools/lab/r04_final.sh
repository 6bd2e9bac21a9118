#!/bin/bash
# Round-4 closing run on the GPU box: the whole -m gpu suite, smoke(), every profile (tools/lab/r04_profiles.sh), then the A/B tables.
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
O=gpurun_out/r04p
mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
timeout 2400 bash tools/lab/r04_profiles.sh > $O/profiles.log 2>&1
OUT=$O/tok_ab.json ROUNDS=4 timeout 600 python tools/tok_ab.py "" "gemm_sched=81" "attn_vit=3" "attn_vit=6" "tokenize_streams=1" > $O/tok_ab.log 2>&1
timeout 400 python tools/gemm_sustained.py > $O/gemm_sustained.log 2>&1
cp gpurun_out/gemm_sustained.json $O/gemm_sustained.json
OUT=$O/decode_ab.json ROUNDS=5 timeout 600 python tools/decode_ab.py "" "skinny_splitk=0" > $O/decode_ab.log 2>&1
tail -3 $O/pytest_gpu.log; cat $O/smoke.log | tail -2; tail -c 1500 $O/bench_final.json; tail -5 $O/decode_ab.log
echo done
