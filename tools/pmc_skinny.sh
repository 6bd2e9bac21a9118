#!/bin/bash
# HBM traffic of the decode GEMM (SEED-LLaMA-8B, batch 32): gpurun_out/pmc_skinny_summary.json
#   tools/pmc_skinny.sh                      q/k/v shape (12288 x 4096, uncut split-K kernel)
#   tools/pmc_skinny.sh 22016 4096 swiglu    gate/up (cut split-K kernel) -> gpurun_out/pmc_skinny_22016_summary.json
R=${GRAFT_REPO_ROOT:-$PWD}
N=${1:-12288}; K=${2:-4096}; FORM=${3:-}
cd /tmp && export TMPDIR=/tmp
i=0
for P in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $P --kernel-trace --output-format csv -d /tmp/pmcs_p$i -- python $R/tools/skinny_one.py $N $K 32 $FORM > /tmp/pmcs_p$i.log 2>&1
done
cd $R
N=$N K=$K python - <<'PY'
import csv, glob, collections, json, os
N, K = int(os.environ["N"]), int(os.environ["K"])
out = {}
for f in glob.glob("/tmp/pmcs_p*/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "gemm_skinny" in r.get("Kernel_Name", ""):
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    out.update({k: sum(v) / len(v) for k, v in acc.items()})
out["traffic_bytes"] = (2 * out.get("FETCH_SIZE", 0) + out.get("WRITE_SIZE", 0)) * 1024
out["algorithmic_bytes"] = N * K * 2 + 32 * K * 2 + 32 * N * 2
json.dump(out, open("gpurun_out/pmc_skinny_summary.json" if N == 12288 else "gpurun_out/pmc_skinny_%d_summary.json" % N, "w"), indent=1)
print(json.dumps(out, indent=1))
PY
