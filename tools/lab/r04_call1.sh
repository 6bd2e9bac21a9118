#!/bin/bash
# round 4, call 1: the new parity / boundary tests of the round and the default bench line
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
O=gpurun_out/r04c1
mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x -s -k "14b or peaked or live_reference or fork_join or sticky or padded or end_to_end_on_device or full_depth_config3" > $O/pytest_new.log 2>&1; echo "pytest rc=$?" >> $O/pytest_new.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/bench.err
grep -E "passed|failed|error|rc=" $O/pytest_new.log | tail -5
tail -c 3000 $O/bench.json; tail -3 $O/bench.err
