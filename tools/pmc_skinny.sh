#!/bin/bash
# HBM traffic of the decode GEMM (QKV shape of SEED-LLaMA-8B, batch 32): gpurun_out/pmc_skinny_summary.json
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
i=0
for P in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $P --kernel-trace --output-format csv -d /tmp/pmcs_p$i -- python $R/tools/skinny_one.py 12288 4096 32 > /tmp/pmcs_p$i.log 2>&1
done
cd $R
python - <<'PY'
import csv, glob, collections, json
out = {}
for f in glob.glob("/tmp/pmcs_p*/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "gemm_skinny" in r.get("Kernel_Name", ""):
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    out.update({k: sum(v) / len(v) for k, v in acc.items()})
out["traffic_bytes"] = (2 * out.get("FETCH_SIZE", 0) + out.get("WRITE_SIZE", 0)) * 1024
out["algorithmic_bytes"] = 12288 * 4096 * 2 + 32 * 4096 * 2 + 32 * 12288 * 2
json.dump(out, open("gpurun_out/pmc_skinny_summary.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
