#!/bin/bash
# round 4, call 30: sanity of the clean rebuild at HEAD: smoke(), attention + tokenizer tests, a short bench line
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
O=gpurun_out/r04c30
mkdir -p $O
export TMPDIR=/tmp
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_tokenizer.py -m gpu -q -x -k "attention or tokenize or full_size or layernorm_fold" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
timeout 600 python bench.py --no-cpu-baseline --no-prefill14b --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/bench.err
tail -2 $O/smoke.log; tail -3 $O/pytest.log; tail -c 600 $O/bench.json | head -c 600; tail -1 $O/bench.err
