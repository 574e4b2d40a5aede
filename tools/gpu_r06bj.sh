#!/bin/bash
# round 6, visit bj: same-address atomics at the end of streaming kernels -- grid caps
TAG=${1:-r06bj}
mkdir -p gpurun_out
export TAG
. tools/gpu_step.sh
run() { echo "--- $*" >> gpurun_out/${TAG}_atomics.txt; env "$@" timeout -k 10 200 python tools/bench_atomics.py >> gpurun_out/${TAG}_atomics.txt 2>&1; }
step a1 run S2T_BYPASS_UP_BWD16=0 S2T_NB_BWD_BLOCKS=1024
step a2 run S2T_BUP_BLOCKS=2048 S2T_NB_BWD_BLOCKS=512
step a3 run S2T_BUP_BLOCKS=1024 S2T_NB_BWD_BLOCKS=256
step a4 run S2T_BUP_BLOCKS=512 S2T_NB_BWD_BLOCKS=128
step a5 run S2T_BUP_BLOCKS=256 S2T_NB_BWD_BLOCKS=64
cat gpurun_out/${TAG}_atomics.txt
