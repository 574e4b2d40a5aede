#!/bin/bash
# round 6, visit bq: conv3x3_c1 weight gradient, workgroup count (kernel time from a trace of the step)
TAG=${1:-r06bq}
mkdir -p gpurun_out
export TAG
. tools/gpu_step.sh
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
for nb in ${NBS:-2048 4096}; do
  rm -rf gpurun_out/${TAG}_prof
  export S2T_C1_WGRAD_BLOCKS=$nb
  step prof_$nb bash -c "rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_prof -o t --output-format csv -- python bench.py --steps 5 --warmup 4 --no-cpu-baseline --profile-steps 0 > /dev/null 2> gpurun_out/${TAG}_err.txt; grep -h 'conv3x3_c1_wgrad' \$(find gpurun_out/${TAG}_prof -name '*kernel_stats.csv') | cut -c1-60,100-200 | sed 's/^/nb=$nb /' | tee -a gpurun_out/${TAG}_c1.txt"
  rm -rf gpurun_out/${TAG}_prof
done
