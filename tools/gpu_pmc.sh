#!/bin/bash
# Two separate PMC passes (FETCH_SIZE, WRITE_SIZE) over a short C3 bench run; kernel-trace only.
set -o pipefail
TAG=${1:-r02}
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_${TAG}_$C
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d gpurun_out/pmc_${TAG}_$C -o pmc -- python bench.py --steps 2 --warmup 2 --no-cpu-baseline --profile-steps 0 > gpurun_out/pmc_${TAG}_$C.json 2> gpurun_out/pmc_${TAG}_$C.err
  echo "$C rc=$?"
  find gpurun_out/pmc_${TAG}_$C -name "*kernel_trace.csv" -delete
done
python tools/pmc_traffic.py gpurun_out/pmc_${TAG}_FETCH_SIZE gpurun_out/pmc_${TAG}_WRITE_SIZE gpurun_out/roofline_traffic_${TAG}.json | head -30
# the raw per-dispatch tables are large: keep only the summary
rm -rf gpurun_out/pmc_${TAG}_FETCH_SIZE gpurun_out/pmc_${TAG}_WRITE_SIZE
