#!/bin/bash
# round 6, visit k: how the step ends (side stream's frontend weight gradients vs the optimizer)
TAG=${1:-r06k}
mkdir -p gpurun_out
export TAG
. tools/gpu_step.sh
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
rm -rf gpurun_out/${TAG}_prof
step prof bash -c "rocprofv3 --kernel-trace -d gpurun_out/${TAG}_prof -o t --output-format csv -- python bench.py --steps 5 --warmup 4 --no-cpu-baseline --profile-steps 0 > gpurun_out/${TAG}_profbench.json 2> gpurun_out/${TAG}_profbench.err; tail -c 300 gpurun_out/${TAG}_profbench.json"
T=$(find gpurun_out/${TAG}_prof -name "*_kernel_trace.csv")
python tools/prof_tail.py $T 3 | tee gpurun_out/${TAG}_tail.txt
rm -rf gpurun_out/${TAG}_prof
