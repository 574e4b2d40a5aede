#!/bin/bash
# round 6, visit ap: stream-placement toggles re-measured on the tree with the light weight-gradient kernels
TAG=${1:-r06ap}
mkdir -p gpurun_out
export TAG
. tools/gpu_step.sh
b() {
  local name=$1; local cfg=$2; shift; shift
  step bench_$name bash -c "$* timeout -k 10 600 python bench.py --config $cfg --steps 20 --warmup 5 --no-cpu-baseline --profile-steps 0 2>> gpurun_out/${TAG}_bench.err | tail -1 | python -c \"import json,sys; d=json.loads(sys.stdin.read()); print('$name', round(d['ms_per_step'],2))\" | tee -a gpurun_out/${TAG}_ab.txt"
}
for r in 1 2; do
b base_$r C3
b front0_$r C3 S2T_FRONT_W_SIDE=0
b front3_$r C3 S2T_FRONT_W_SIDE=3
b front4_$r C3 S2T_FRONT_W_SIDE=4
b convw0_$r C3 S2T_CONV_W_SIDE=0
b whiten_main_$r C3 S2T_WHITEN_STREAM=0
b margin1_$r C3 S2T_X3P_MARGIN=1.0
done
