#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
P2="SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM SQ_WAVES"
P3="SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_FLAT SQ_INSTS_SMEM GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $R/gpurun_out/pmc_attn_p$i -- python $R/tools/attn_bench.py > $R/gpurun_out/pmc_attn_p$i.log 2>&1
done
cd $R
python - <<'PY'
import csv, glob, collections, json, os
out = {}
for d in sorted(glob.glob("gpurun_out/pmc_attn_p*")):
    if not os.path.isdir(d): continue
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r.get("Kernel_Name", "")
            if "attn" in k:
                name = "vit" if "attn_vit" in k else ("trv" if "Lb1EEEv" in k.replace("ELb1ELb1EEE","ELb1ELb1EEE") and k.rstrip().endswith("true>") else k[-40:])
                acc[k[:60] + "|" + k[-30:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            out.setdefault(k, {}).update({c: sum(x) / len(x) for c, x in v.items()})
json.dump(out, open("gpurun_out/pmc_attn_summary.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
