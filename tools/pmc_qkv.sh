#!/bin/bash
# HBM traffic + L2 hit counters of the shipped QKV GEMM (run on the GPU box through gpurun): gpurun_out/pmc_qkv_summary.json
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
i=0
for P in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $P --kernel-trace --output-format csv -d /tmp/pmcq_p$i -- python $R/tools/gemm_one.py 256 65792 4224 1408 3 > /tmp/pmcq_p$i.log 2>&1
done
cd $R
python - <<'PY'
import csv, glob, collections, json
out = {}
for f in glob.glob("/tmp/pmcq_p*/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "gemm" in r.get("Kernel_Name", ""):
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    out.update({k: sum(v) / len(v) for k, v in acc.items()})
out["traffic_bytes"] = (2 * out.get("FETCH_SIZE", 0) + out.get("WRITE_SIZE", 0)) * 1024
# effective clock and MFMA-busy share of the launch (the pass that carried GRBM_GUI_ACTIVE: its own kernel-trace durations)
durs = []
for f in glob.glob("/tmp/pmcq_p4/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm256" in r.get("Kernel_Name", ""):
            durs.append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
if durs and out.get("GRBM_GUI_ACTIVE"):
    ns = sum(durs) / len(durs)
    out["launch_ns_in_pmc_pass"] = ns
    out["effective_clock_ghz"] = out["GRBM_GUI_ACTIVE"] / 8 / ns                 # (the counter is summed over the 8 XCDs)
    # SQ_VALU_MFMA_BUSY_CYCLES is summed over the SIMDs of the chip in cycles (1024 SIMDs on MI355X): busy share of the matrix pipes
    out["mfma_busy_frac"] = out.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (out["GRBM_GUI_ACTIVE"] / 8 * 1024)
out["algorithmic_bytes"] = 2.0 * (65792 * 1408 + 4224 * 1408 + 65792 * 4224)
out["traffic_over_algorithmic"] = out["traffic_bytes"] / out["algorithmic_bytes"]
json.dump(out, open("gpurun_out/pmc_qkv_summary.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
