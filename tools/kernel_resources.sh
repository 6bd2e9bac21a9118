#!/bin/bash
# Per-kernel register / scratch usage of one HIP source (compiler remarks), one line per kernel.
# usage: tools/kernel_resources.sh seed_amd/csrc/gemm_bf16.hip [extra hipcc flags]
src=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c "$src" -o /dev/null -Rpass-analysis=kernel-resource-usage "$@" 2>&1 |
python3 -c '
import re, sys, subprocess
cur = {}
def flush():
    if cur:
        name = subprocess.run(["c++filt", cur.get("Name", "?")], capture_output=True, text=True).stdout.strip()
        name = re.sub(r"\(anonymous namespace\)::", "", name).split("(")[0]
        print("%-46s vgpr %3s agpr %3s sgpr %3s scratch %4s occupancy %s lds %s" % (name[:46], cur.get("VGPRs"), cur.get("AGPRs"), cur.get("SGPRs"),
              cur.get("ScratchSize [bytes/lane]"), cur.get("Occupancy [waves/SIMD]"), cur.get("LDS Size [bytes/block]")))
for line in sys.stdin:
    m = re.search(r"remark: [^ ]+ +(?:Function )?([A-Za-z\[\]/ ]+): (\S+) \[-Rpass", line) or re.search(r": ([A-Za-z\[\]/ ]+): (\S+) \[-Rpass", line)
    if not m:
        continue
    k, v = m.group(1).strip(), m.group(2)
    if k.endswith("Name"):
        flush(); cur = {"Name": v}
    else:
        cur[k] = v
flush()
'
