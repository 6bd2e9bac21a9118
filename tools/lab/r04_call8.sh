#!/bin/bash
# round 4, call 8: PMC of the fc1 launch (N = 6144, K = 1408, LayerNorm fold) with the GELU table epilogue and with the plain BIAS epilogue:
# what the activation costs in LDS cycles / bank conflicts / VALU instructions
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/r04c8
mkdir -p $O
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
P2="SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
for epi in 1 2; do
  i=0
  for P in "$P1" "$P2"; do
    i=$((i+1))
    EPI=$epi timeout 120 rocprofv3 --pmc $P --kernel-trace --output-format csv -d /tmp/pmcf_e${epi}_p$i -- python $R/tools/gemm_one.py 256 65792 6144 1408 3 > $O/e${epi}_p$i.log 2>&1
  done
done
cd $R
python - <<'PY'
import csv, glob, collections, json
out = {}
for epi in (1, 2):
    acc = collections.defaultdict(list)
    for f in glob.glob(f"/tmp/pmcf_e{epi}_p*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "gemm256" in r.get("Kernel_Name", ""):
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    out["BIAS" if epi == 1 else "BIAS_GELU"] = {k: sum(v) / len(v) for k, v in acc.items()}
json.dump(out, open("gpurun_out/r04c8/pmc_fc1_epilogues.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
