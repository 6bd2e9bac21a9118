#!/bin/bash
# Traffic past the L2s + L2 hit counters of the ViT attention at B = 128: lock-step kernel, staggered kernel with the plain walk, staggered
# kernel with the XCD-aware walk (run on the GPU box through gpurun): gpurun_out/pmc_attn_traffic.json
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
for arm in "3 1" "5 0" "5 1"; do
  set -- $arm
  i=0
  for P in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1))
    timeout 120 rocprofv3 --pmc $P --kernel-trace --output-format csv -d /tmp/pmca_$1_$2_p$i -- python $R/tools/attn_one.py 128 $1 $2 3 > /tmp/pmca_$1_$2_p$i.log 2>&1
  done
done
cd $R
python - <<'PY'
import csv, glob, collections, json
res = {}
for arm, name in (("3_1", "lock-step kernel (attn_vit=3)"), ("5_0", "staggered kernel, plain walk (attn_vit=5, attn_xcd=0)"), ("5_1", "staggered kernel, XCD-aware walk (attn_vit=5, attn_xcd=1)")):
    out = {}
    for f in glob.glob(f"/tmp/pmca_{arm}_p*/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "attn_vit" in r.get("Kernel_Name", ""):
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        out.update({k: sum(v) / len(v) for k, v in acc.items()})
    out["traffic_bytes"] = (2 * out.get("FETCH_SIZE", 0) + out.get("WRITE_SIZE", 0)) * 1024      # same unit corrections as tools/pmc_qkv.sh
    out["algorithmic_bytes"] = 128 * 257 * 1408 * 2 * 4
    out["traffic_over_algorithmic"] = out["traffic_bytes"] / out["algorithmic_bytes"]
    if out.get("TCC_HIT_sum"): out["l2_hit_rate"] = out["TCC_HIT_sum"] / (out["TCC_HIT_sum"] + out["TCC_MISS_sum"])
    res[name] = out
json.dump(res, open("gpurun_out/pmc_attn_traffic.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
