#!/bin/bash
# PMC counters of the GEMM variants on one shape (inside gpurun): bash tools/x3p_pmc.sh TAG M N K
set -o pipefail
TAG=$1; M=$2; NN=$3; K=$4
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rocprofv3 -L > gpurun_out/counters_list.txt 2>&1
grep -o "SQ_[A-Z_0-9]*\|GRBM_[A-Z_0-9]*\|TCP_[A-Z_0-9]*\|TCC_[A-Z_0-9]*" gpurun_out/counters_list.txt | sort -u > gpurun_out/counters_names.txt
wc -l gpurun_out/counters_names.txt
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_LDS"
P2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE"
i=0
for P in "$P1" "$P2"; do
  i=$((i+1))
  rm -rf gpurun_out/${TAG}_pmc$i
  rocprofv3 --pmc $P --kernel-trace --output-format csv -d gpurun_out/${TAG}_pmc$i -o pmc -- python tools/x3p_one.py $M $NN $K 6 > gpurun_out/${TAG}_pmc$i.log 2>&1
  echo "pass $i rc=$?"; tail -2 gpurun_out/${TAG}_pmc$i.log
done
python - <<PY
import csv, glob, collections
for i in (1, 2):
    fs = glob.glob("gpurun_out/${TAG}_pmc%d/**/*counter_collection.csv" % i, recursive=True)
    if not fs:
        print("no counters for pass", i); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        agg[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in agg.items():
        if "gemm" not in k and "x3p" not in k and "Cijk" not in k:
            continue
        print(k)
        print("   " + "  ".join("%s=%.3g" % (c, sum(v) / len(v)) for c, v in sorted(cs.items())))
PY
rm -rf gpurun_out/${TAG}_prof
rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_prof -o st --output-format csv -- python tools/x3p_one.py $M $NN $K 10 > /dev/null 2>&1
python - <<PY
import csv, glob
f = glob.glob("gpurun_out/${TAG}_prof/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:12]:
    print(f"{float(r['AverageNs'])/1e3:9.1f} us x{r['Calls']:>5}  {r['Name'][:100]}")
PY
find gpurun_out -name "*_kernel_trace.csv" -delete
find gpurun_out -name "*counter_collection.csv" -size +20M -delete
