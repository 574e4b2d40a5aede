#!/bin/bash
# round 6, visit w: ungrouped weight gradients on the all-waves-split 128 x 128 / 128 x 192 form (grouped stay on the W form)
TAG=${1:-r06w}
mkdir -p gpurun_out
export TAG
. tools/gpu_step.sh
b() {
  local name=$1; shift
  step bench_$name bash -c "$* timeout -k 10 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --profile-steps 0 2>> gpurun_out/${TAG}_bench.err | tail -1 | python -c \"import json,sys; d=json.loads(sys.stdin.read()); print('$name', round(d['ms_per_step'],2))\" | tee -a gpurun_out/${TAG}_ab.txt"
}
for r in 1 2; do
b default$r
b tile22_$r S2T_TN_TILE=22
b tile23_$r S2T_TN_TILE=23
b tile22_b1024_$r S2T_TN_TILE=22 S2T_TN_BLOCKS=1024
b tile22_b512_$r S2T_TN_TILE=22 S2T_TN_BLOCKS=512
b tile12_$r S2T_TN_TILE=12
done
