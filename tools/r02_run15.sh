#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out/r02
cd /tmp && export TMPDIR=/tmp
for cfg in "tokenize_streams=2" "tokenize_streams=1"; do
  tag=$(echo $cfg | tr -c 'a-z0-9' '_')
  rm -rf /tmp/tr_$tag
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$tag -- python $R/tools/tok_trace.py "$cfg" > $R/gpurun_out/r02/trace_$tag.log 2>&1
  f=$(ls /tmp/tr_$tag/*/*kernel_trace.csv | head -1)
  python $R/tools/kernel_timeline.py $f /tmp/kt_$tag.csv >> $R/gpurun_out/r02/trace_$tag.log 2>&1
  python $R/tools/timeline_stats.py /tmp/kt_$tag.csv > $R/gpurun_out/r02/timeline_$tag.txt 2>&1
  tail -n 3000 /tmp/kt_$tag.csv > $R/gpurun_out/r02/kt_$tag.csv
done
echo done
