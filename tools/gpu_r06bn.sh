#!/bin/bash
# round 6, visit bn: norm_bypass_bwd16 rows per wave
TAG=${1:-r06bn}
mkdir -p gpurun_out
export TAG
. tools/gpu_step.sh
run() { echo "--- $*" >> gpurun_out/${TAG}_atomics.txt; env "$@" timeout -k 10 200 python tools/bench_atomics.py 2>&1 | grep norm_bypass >> gpurun_out/${TAG}_atomics.txt; }
step a1 run S2T_NB_BWD_RPW=8
step a2 run S2T_NB_BWD_RPW=12
step a3 run S2T_NB_BWD_RPW=16
step a4 run S2T_NB_BWD_RPW=24
step a5 run S2T_NB_BWD_RPW=32
cat gpurun_out/${TAG}_atomics.txt
