#!/bin/bash
# round 6, visit u: the frontend's weight-gradient products on the TN forms, per tunable
TAG=${1:-r06u}
mkdir -p gpurun_out
export TAG
. tools/gpu_step.sh
v() { step "v_$1" bash -c "TAGV='$*' $* timeout -k 10 200 python tools/debug/tn_front.py 2>&1 | tail -1 | tee -a gpurun_out/${TAG}_tn.txt"; }
v S2T_X=0
v S2T_TN_BLOCKS=256
v S2T_TN_BLOCKS=1024
v S2T_TN_BLOCKS=2048
v S2T_TN_W=0
v S2T_TN_W=0 S2T_TN_BLOCKS=3072
v S2T_TN_W=0 S2T_TN_TILE=22
v S2T_GEMM_ARITH_W=3
