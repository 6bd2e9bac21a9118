#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd $R
mkdir -p gpurun_out/r03
export TMPDIR=/tmp
{
echo "=== decode A/B (interleaved graphs)"; date
OUT=gpurun_out/r03/decode_ab_call17.json timeout 900 python tools/decode_ab.py "" "skinny_splitk=3" "skinny_splitk=0" "decode_attn_early=1" "decode_attn_early=2" "skinny_splitk=2" 2>&1 | tail -8
echo "=== decode-only kernel stats"; date
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03/decode_prof -- python $R/tools/decode_only.py > $R/gpurun_out/r03/decode_prof.log 2>&1)
find gpurun_out/r03/decode_prof -name '*kernel_stats.csv' -exec cp {} gpurun_out/r03/decode_kernel_stats.csv \;
find gpurun_out/r03/decode_prof -name '*kernel_trace.csv' -exec cp {} gpurun_out/r03/decode_kernel_trace.csv \;
rm -rf gpurun_out/r03/decode_prof
tail -2 gpurun_out/r03/decode_prof.log
python - <<'PY'
import csv
rows = list(csv.DictReader(open("gpurun_out/r03/decode_kernel_stats.csv")))
for r in rows[:12]:
    print(r["Name"][:90].ljust(90), r["Calls"].rjust(6), ("%.1f" % (float(r["AverageNs"]) / 1e3)).rjust(8), "us", r["Percentage"].rjust(7))
# gaps between consecutive kernels inside the replayed steps (last 2000 kernels of the trace)
tr = list(csv.DictReader(open("gpurun_out/r03/decode_kernel_trace.csv")))
tr.sort(key=lambda r: int(r["Start_Timestamp"]))
tail = tr[-2000:]
gaps = [int(b["Start_Timestamp"]) - int(a["End_Timestamp"]) for a, b in zip(tail, tail[1:])]
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in tail)
span = int(tail[-1]["End_Timestamp"]) - int(tail[0]["Start_Timestamp"])
gaps.sort()
print("last 2000 kernels: span %.1f us, in kernels %.1f us (%.1f %%), median gap %.2f us, mean gap %.2f us" % (span / 1e3, busy / 1e3, 100.0 * busy / span, gaps[len(gaps) // 2] / 1e3, sum(gaps) / len(gaps) / 1e3))
PY
rm -f gpurun_out/r03/decode_kernel_trace.csv
date
} > gpurun_out/r03/call17.log 2>&1
tail -60 gpurun_out/r03/call17.log
