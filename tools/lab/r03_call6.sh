#!/bin/bash
# round 3, GPU call 6: tile-group sweep under the new schedule, in-path stream-K, kernel timelines (two streams / one stream)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
{
echo "=== group sweep"; date
timeout 300 python tools/gemm_group_sweep.py 2>&1 | tail -12
echo "=== end-to-end A/B"; date
ROUNDS=4 OUT=gpurun_out/r03/tok_ab_call6.json timeout 600 python tools/tok_ab.py "" "tokenize_streamk=1" "gemm_group_m=4" "gemm_group_m=8" 2>&1 | python -c "
import sys, json
t = sys.stdin.read()
try:
    d = json.loads(t[t.index('{\n'):])
    for k, v in d.items(): print(repr(k), v['median_ms'], v['img_s'], v['all_ms'])
except Exception as e:
    print(t[-3000:])
"
for st in 2 1; do
echo "=== kernel timeline, tokenize_streams=$st"; date
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt$st -- python $R/tools/tok_trace.py tokenize_streams=$st > /tmp/kt$st.log 2>&1)
f=$(find /tmp/kt$st -name '*kernel_trace.csv' | head -1)
python tools/kernel_timeline.py $f gpurun_out/r03/kt_compact_$st.csv && python tools/timeline_stats.py gpurun_out/r03/kt_compact_$st.csv | tee gpurun_out/r03/timeline_streams_$st.txt
rm -f gpurun_out/r03/kt_compact_$st.csv
done
date
} > gpurun_out/r03/call6.log 2>&1
tail -90 gpurun_out/r03/call6.log
