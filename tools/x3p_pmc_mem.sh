#!/bin/bash
# Memory-path PMC counters of the x3p GEMM (and hipBLASLt beside it) on one shape, inside gpurun:
#   bash tools/x3p_pmc_mem.sh TAG M N K
set -o pipefail
TAG=$1; M=$2; NN=$3; K=$4
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
PASS=(
 "SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_LEVEL_VMEM SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
 "TCP_TCC_READ_REQ_LATENCY TCP_TCC_READ_REQ TCP_TCC_WRITE_REQ_LATENCY TCP_TCC_WRITE_REQ TCP_PENDING_STALL_CYCLES TCP_TCR_TCP_STALL_CYCLES TCP_GATE_EN1 TCP_TOTAL_CACHE_ACCESSES"
 "TCP_TA_TCP_STATE_READ TCP_TCP_TA_DATA_STALL_CYCLES TCP_TD_TCP_STALL_CYCLES TCP_LFIFO_STALL_CYCLES TCP_RFIFO_STALL_CYCLES TCP_READ_TAGCONFLICT_STALL_CYCLES TCP_TOTAL_READ TCP_TOTAL_WRITE"
 "TCC_HIT TCC_MISS TCC_REQ TCC_BUSY TCC_CYCLE TCC_TAG_STALL TCC_EA0_WRREQ_STALL TCC_TOO_MANY_EA_WRREQS_STALL"
 "TCC_EA0_RDREQ TCC_EA0_WRREQ TCC_SRC_FIFO_FULL TCC_LATENCY_FIFO_FULL TCC_IB_STALL TCC_EA0_RDREQ_LEVEL TCC_EA0_WRREQ_LEVEL TCC_WRITEBACK"
)
i=0
for P in "${PASS[@]}"; do
  i=$((i+1))
  rm -rf gpurun_out/${TAG}_m$i
  rocprofv3 --pmc $P --kernel-trace --output-format csv -d gpurun_out/${TAG}_m$i -o pmc -- python tools/x3p_one.py $M $NN $K 4 > gpurun_out/${TAG}_m$i.log 2>&1
  echo "pass $i rc=$?"
done
python - <<PY
import csv, glob, collections
for i in range(1, 6):
    fs = glob.glob("gpurun_out/${TAG}_m%d/**/*counter_collection.csv" % i, recursive=True)
    if not fs:
        print("no counters for pass", i); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        agg[r["Kernel_Name"][:64]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in agg.items():
        if "x3p_db" not in k and "Cijk" not in k:
            continue
        print(k)
        print("   " + "  ".join("%s=%.4g" % (c, sum(v) / len(v)) for c, v in sorted(cs.items())))
PY
find gpurun_out -name "*_kernel_trace.csv" -delete
find gpurun_out -name "*counter_collection.csv" -size +20M -delete
