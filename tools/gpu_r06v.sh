#!/bin/bash
# round 6, visit v: in-step A/B of the weight-gradient kernels' grid / form tunables
TAG=${1:-r06v}
mkdir -p gpurun_out
export TAG
. tools/gpu_step.sh
b() {
  local name=$1; shift
  step bench_$name bash -c "$* timeout -k 10 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --profile-steps 0 2>> gpurun_out/${TAG}_bench.err | tail -1 | python -c \"import json,sys; d=json.loads(sys.stdin.read()); print('$name', round(d['ms_per_step'],2))\" | tee -a gpurun_out/${TAG}_ab.txt"
}
for r in 1 2; do
b default$r
b tnblocks1024_$r S2T_TN_BLOCKS=1024
b wblocks1024_$r S2T_TN_W_BLOCKS=1024
b wblocks768_$r S2T_TN_W_BLOCKS=768
b now_tile22_$r S2T_TN_W=0 S2T_TN_TILE=22
b now_$r S2T_TN_W=0
done
