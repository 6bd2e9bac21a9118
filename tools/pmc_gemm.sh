#!/bin/bash
# PMC passes for the two GEMM kernels (run on the GPU box through gpurun). Output: gpurun_out/pmc_<tag>_<pass>/
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o -E "\b(SQ_[A-Z_0-9]+|TCC_[A-Z_0-9a-z]+|FETCH_SIZE|WRITE_SIZE|GRBM_[A-Z_]+|TCP_[A-Z_0-9a-z]+|MfmaUtil|VALUBusy)\b" | sort -u > $R/gpurun_out/counters.txt
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
P2="SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_UNALIGNED_STALL SQ_INSTS_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE SQ_WAVES"
P3="FETCH_SIZE"
P4="WRITE_SIZE"
P5="TCC_HIT_sum TCC_MISS_sum"
for cfg in "256 65792 4224 1408" "256 65792 1408 6144"; do
  set -- $cfg; tag="v$1_N$3_K$4"
  i=0
  for P in "$P1" "$P2" "$P3" "$P4" "$P5"; do
    i=$((i+1))
    timeout 120 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${tag}_p$i -- python $R/tools/gemm_one.py $cfg 3 > $R/gpurun_out/pmc_${tag}_p$i.log 2>&1
  done
done
cd $R
python - <<'PY'
import csv, glob, collections, json, os
out = {}
for d in sorted(glob.glob("gpurun_out/pmc_*_p*")):
    if not os.path.isdir(d): continue
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "gemm" in r.get("Kernel_Name", ""):
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        out[os.path.basename(d)] = {k: sum(v) / len(v) for k, v in acc.items()}
json.dump(out, open("gpurun_out/pmc_summary.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
