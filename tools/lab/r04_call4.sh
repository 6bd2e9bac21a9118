#!/bin/bash
# round 4, call 4: stream-K tail for the 14B prefill's MFMA GEMMs (A/B), then the whole -m gpu suite and smoke()
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
O=gpurun_out/r04c4
mkdir -p $O
export TMPDIR=/tmp
AB="prefill_streamk=0|prefill_streamk=1" ROUNDS=5 timeout 600 python tools/prefill_bench.py > $O/prefill_ab.log 2>&1; echo "prefill rc=$?" >> $O/prefill_ab.log
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
tail -4 $O/prefill_ab.log; tail -4 $O/pytest_gpu.log; tail -3 $O/smoke.log
