#!/bin/bash
# FETCH_SIZE / WRITE_SIZE calibration probe + the two PMC passes over a short bench run.
# usage (inside gpurun): bash tools/gpu_pmc2.sh TAG [CONFIG]
set -o pipefail
TAG=${1:-r03}; CFG=${2:-C3}
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
if [ ! -x tools/probes/fetch_probe ]; then
  hipcc -O3 --offload-arch=gfx950 tools/probes/fetch_probe.hip -o tools/probes/fetch_probe || { echo "fetch_probe build failed"; exit 1; }
fi
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/cal_$C
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d gpurun_out/cal_$C -o cal -- tools/probes/fetch_probe > gpurun_out/cal_$C.log 2>&1
  rc=$?; echo "cal $C rc=$rc"
  if [ $rc -ne 0 ]; then tail -5 gpurun_out/cal_$C.log; exit 1; fi
done
python - <<PY > gpurun_out/${TAG}_fetch_calibration.txt
import csv, glob, collections
print("# tools/probes/fetch_probe.hip: each kernel streams 512 MiB once (read_rows64: 128 MiB); counters in KiB")
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("gpurun_out/cal_%s/**/*counter_collection.csv" % C, recursive=True)[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == C:
            agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        exp = (128 if "rows64" in k else 512) * 1024.0
        print("%-11s %-60s mean %12.0f KiB  = %.3f of the bytes streamed" % (C, k[:60], sum(v) / len(v), sum(v) / len(v) / exp))
PY
cat gpurun_out/${TAG}_fetch_calibration.txt
rm -rf gpurun_out/cal_FETCH_SIZE gpurun_out/cal_WRITE_SIZE
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_${TAG}_$C
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d gpurun_out/pmc_${TAG}_$C -o pmc -- python3 bench.py --config $CFG --steps 2 --warmup 5 --sample-every 1 --roofline-kernel s2t_gemm_x3p --no-cpu-baseline --profile-steps 0 > gpurun_out/pmc_${TAG}_$C.json 2> gpurun_out/pmc_${TAG}_$C.err
  echo "$C rc=$?"
  find gpurun_out/pmc_${TAG}_$C -name "*kernel_trace.csv" -delete
done
python tools/pmc_traffic.py gpurun_out/pmc_${TAG}_FETCH_SIZE gpurun_out/pmc_${TAG}_WRITE_SIZE gpurun_out/roofline_traffic_${TAG}_${CFG}.json gpurun_out/pmc_${TAG}_WRITE_SIZE.json | head -40
rm -rf gpurun_out/pmc_${TAG}_FETCH_SIZE gpurun_out/pmc_${TAG}_WRITE_SIZE
