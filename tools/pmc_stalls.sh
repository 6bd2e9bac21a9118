#!/bin/bash
# Stall-reason attribution of one kernel from SQ / TA / TCP counters ("which instruction class waits on what").
#   bash tools/pmc_stalls.sh                                   # the shipped ViT QKV GEMM launch (gemm_one.py 256 65792 4224 1408)
#   KERNEL=gemm_skinny CMD="python tools/skinny_one.py ..." OUT=pmc_stalls_skinny.json bash tools/pmc_stalls.sh
# Every counter group is its own rocprofv3 pass (--pmc with --kernel-trace only: the form gpurun accepts).  SQ_* cycle counters are in
# quad-cycles summed over waves (SQ_WAVE_CYCLES, SQ_WAIT_*, SQ_ACTIVE_INST_*), SQ_BUSY_CYCLES per SE; the summary divides by SQ_WAVE_CYCLES.
R=${GRAFT_REPO_ROOT:-$PWD}
KERNEL=${KERNEL:-gemm256}
CMD=${CMD:-"python $R/tools/gemm_one.py 256 65792 4224 1408 3"}
OUT=${OUT:-pmc_stalls_qkv.json}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmcs_p*
i=0
for P in \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
  "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS" \
  "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU" \
  "SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_IFETCH SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" \
  "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_BUSY_avr" \
  "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum" \
  "TCC_EA0_WRREQ_DRAM_sum TCC_EA0_WRREQ_64B_sum TCC_REQ_sum TCC_TAG_STALL_sum" \
  "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_TCP_LATENCY_sum" \
  "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_LEVEL_sum TCC_BUBBLE_sum" \
  "TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_WRREQ_WRITE_DRAM_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum"; do
  i=$((i+1))
  timeout 180 rocprofv3 --pmc $P --kernel-trace --output-format csv -d /tmp/pmcs_p$i -- $CMD > /tmp/pmcs_p$i.log 2>&1 || { echo "pass $i ($P) failed:"; tail -3 /tmp/pmcs_p$i.log; }
done
cd $R
KERNEL=$KERNEL OUT=$OUT python - <<'PY'
import csv, glob, collections, json, os
kern, out = os.environ["KERNEL"], {}
for f in sorted(glob.glob("/tmp/pmcs_p*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if kern in r.get("Kernel_Name", ""):
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    out.update({k: sum(v) / len(v) for k, v in acc.items()})
durs = []
for f in glob.glob("/tmp/pmcs_p4/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if kern in r.get("Kernel_Name", ""):
            durs.append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
d = {"kernel": kern, "counters": out}
wc = out.get("SQ_WAVE_CYCLES")
if wc:
    d["share_of_wave_cycles"] = {k: round(out[k] / wc, 4) for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_VALU",
                                                                     "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_MISC",
                                                                     "SQ_INST_CYCLES_VMEM_RD", "SQ_INST_CYCLES_VMEM_WR") if k in out}
if durs and out.get("GRBM_GUI_ACTIVE"):
    ns = sum(durs) / len(durs)
    clk = out["GRBM_GUI_ACTIVE"] / 8 / ns
    d.update(launch_ns_in_pmc_pass=ns, effective_clock_ghz=clk, mfma_busy_frac=out.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (out["GRBM_GUI_ACTIVE"] / 8 * 1024),
             mfma_valu_coexec_frac=out.get("SQ_VALU_MFMA_COEXEC_CYCLES", 0) / (out["GRBM_GUI_ACTIVE"] / 8 * 1024))
if out.get("TCP_TCC_READ_REQ_sum"):
    d["avg_l2_read_round_trip_cycles"] = out.get("TCP_TCC_READ_REQ_LATENCY_sum", 0) / out["TCP_TCC_READ_REQ_sum"]
if out.get("TCC_HIT_sum") is not None and out.get("TCC_MISS_sum") is not None and (out["TCC_HIT_sum"] + out["TCC_MISS_sum"]) > 0:
    d["l2_hit_rate"] = out["TCC_HIT_sum"] / (out["TCC_HIT_sum"] + out["TCC_MISS_sum"])
if out.get("TCC_EA0_RDREQ_sum"):
    d["ea_read_requests_to_dram_share"] = out.get("TCC_EA0_RDREQ_DRAM_sum", 0) / out["TCC_EA0_RDREQ_sum"]
    if out.get("TCC_EA0_RDREQ_128B_sum") is not None:       # exact: requests by size (the 64 B per request of FETCH_SIZE undercounts 128-byte requests)
        n128, n64 = out.get("TCC_EA0_RDREQ_128B_sum", 0), out.get("TCC_EA0_RDREQ_64B_sum", 0)
        n32 = out.get("TCC_EA0_RDREQ_32B_sum", 0)
        d["ea_read_bytes"] = 128 * n128 + 64 * n64 + 32 * n32 + 64 * max(out["TCC_EA0_RDREQ_sum"] - n128 - n64 - n32, 0)
        d["ea_read_requests_by_size"] = {"128B": n128, "64B": n64, "32B": n32, "all": out["TCC_EA0_RDREQ_sum"]}
    else:
        d["ea_read_bytes"] = (out["TCC_EA0_RDREQ_sum"] - out.get("TCC_EA0_RDREQ_32B_sum", 0)) * 64 + out.get("TCC_EA0_RDREQ_32B_sum", 0) * 32
    d["ea_write_bytes"] = 64 * out.get("TCC_EA0_WRREQ_64B_sum", 0) + 32 * max(out.get("TCC_EA0_WRREQ_sum", 0) - out.get("TCC_EA0_WRREQ_64B_sum", 0), 0)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(d, open("gpurun_out/" + os.environ["OUT"], "w"), indent=1)
print(json.dumps(d, indent=1))
PY
