#!/bin/bash
# round 6, visit bp: downsample_bwd grid caps
TAG=${1:-r06bp}
mkdir -p gpurun_out
export TAG
. tools/gpu_step.sh
run() { echo "--- $*" >> gpurun_out/${TAG}_atomics.txt; env "$@" timeout -k 10 200 python tools/bench_atomics.py 2>&1 | grep downsample >> gpurun_out/${TAG}_atomics.txt; }
step a1 run S2T_DS_BWD_BLOCKS=1024
step a2 run S2T_DS_BWD_BLOCKS=512
step a3 run S2T_DS_BWD_BLOCKS=256
step a4 run S2T_DS_BWD_BLOCKS=128
cat gpurun_out/${TAG}_atomics.txt
