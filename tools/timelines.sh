#!/bin/bash
# Kernel timelines of one tokenize pass (B = 256) with one and with two sub-batch streams: per-kernel totals, concurrency, idle gaps.
#   gpurun ... 'bash tools/lab/run.sh <label> sh:tools/timelines.sh'   ->  gpurun_out/timeline_one_stream.txt, timeline_two_streams.txt
R=${GRAFT_REPO_ROOT:-$PWD}
cd "$R" || exit 1
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
for st in 1 2; do
    rm -rf /tmp/kt$st
    (cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt$st -- python "$R/tools/tok_trace.py" tokenize_streams=$st > /tmp/kt$st.log 2>&1)
    f=$(find /tmp/kt$st -name '*kernel_trace.csv' | head -1)
    name=$([ $st = 1 ] && echo one_stream || echo two_streams)
    python tools/kernel_timeline.py "$f" /tmp/kt_compact_$st.csv && NPARTS=$st python tools/timeline_stats.py /tmp/kt_compact_$st.csv > $O/timeline_$name.txt 2>&1
    rm -f /tmp/kt_compact_$st.csv
done
head -16 $O/timeline_one_stream.txt; echo; head -6 $O/timeline_two_streams.txt; tail -4 $O/timeline_two_streams.txt
