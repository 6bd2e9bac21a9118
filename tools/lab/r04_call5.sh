#!/bin/bash
# round 4, call 5: selective stream-K tail for the 14B prefill (only short GEMMs with a badly filled last round)
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
O=gpurun_out/r04c5
mkdir -p $O
AB="prefill_streamk=0|prefill_streamk=2|prefill_streamk=1" ROUNDS=5 timeout 600 python tools/prefill_bench.py > $O/prefill_ab.log 2>&1; echo "prefill rc=$?" >> $O/prefill_ab.log
tail -5 $O/prefill_ab.log
