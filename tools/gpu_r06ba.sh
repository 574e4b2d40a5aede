#!/bin/bash
# round 6, visit ba: launch-by-launch listing of one step on the last tree + ATen call sites
TAG=${1:-r06ba}
mkdir -p gpurun_out
export TAG
. tools/gpu_step.sh
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
rm -rf gpurun_out/${TAG}_prof
step prof bash -c "rocprofv3 --kernel-trace -d gpurun_out/${TAG}_prof -o t --output-format csv -- python bench.py --steps 5 --warmup 4 --no-cpu-baseline --profile-steps 0 > gpurun_out/${TAG}_profbench.json 2> gpurun_out/${TAG}_profbench.err; tail -c 300 gpurun_out/${TAG}_profbench.json"
T=$(find gpurun_out/${TAG}_prof -name "*_kernel_trace.csv")
python tools/prof_sequence.py $T gpurun_out/${TAG}_sequence.txt
rm -rf gpurun_out/${TAG}_prof
step aten bash -c "timeout -k 10 400 python tools/exp_aten_sites.py > gpurun_out/${TAG}_aten_sites.txt 2>&1; tail -3 gpurun_out/${TAG}_aten_sites.txt"
