#!/bin/bash
# round 6, visit y: grid targets / shapes of the all-waves-split weight-gradient form
TAG=${1:-r06y}
mkdir -p gpurun_out
export TAG
. tools/gpu_step.sh
b() {
  local name=$1; shift
  step bench_$name bash -c "$* timeout -k 10 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --profile-steps 0 2>> gpurun_out/${TAG}_bench.err | tail -1 | python -c \"import json,sys; d=json.loads(sys.stdin.read()); print('$name', round(d['ms_per_step'],2))\" | tee -a gpurun_out/${TAG}_ab.txt"
}
B="S2T_TN_W=0 S2T_TN_GROUP_TILE=22 S2T_TN_TILE=22"
for r in 1 2; do
b base_$r $B
b gb1024_$r $B S2T_TN_GROUP_BLOCKS=1024
b gb2048_$r $B S2T_TN_GROUP_BLOCKS=2048
b ub1024_$r $B S2T_TN_BLOCKS=1024
b ub512_$r $B S2T_TN_BLOCKS=512
b u23_$r S2T_TN_W=0 S2T_TN_GROUP_TILE=22 S2T_TN_TILE=23
b g12_$r S2T_TN_W=0 S2T_TN_GROUP_TILE=12 S2T_TN_TILE=22
done
