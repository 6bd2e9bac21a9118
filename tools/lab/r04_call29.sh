#!/bin/bash
# round 4, call 29: kernel timelines of one tokenize pass at HEAD (one stream and two streams), per-kernel totals
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
O=gpurun_out/r04c29
mkdir -p $O
export TMPDIR=/tmp
for st in 1 2; do
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt$st -- python $R/tools/tok_trace.py tokenize_streams=$st > /tmp/kt$st.log 2>&1)
f=$(find /tmp/kt$st -name '*kernel_trace.csv' | head -1)
python tools/kernel_timeline.py $f $O/kt_compact_$st.csv && NPARTS=$st python tools/timeline_stats.py $O/kt_compact_$st.csv > $O/timeline_streams_$st.txt 2>&1
rm -f $O/kt_compact_$st.csv
done
head -22 $O/timeline_streams_1.txt; echo; head -8 $O/timeline_streams_2.txt; tail -4 $O/timeline_streams_2.txt
