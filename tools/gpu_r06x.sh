#!/bin/bash
# round 6, visit x: grouped weight gradients on the all-waves-split 128 x 128 form (two-piece arithmetic) against the W form
TAG=${1:-r06x}
mkdir -p gpurun_out
export TAG
. tools/gpu_step.sh
step tests bash -c "S2T_TN_W=0 S2T_TN_GROUP_TILE=22 S2T_TN_TILE=22 timeout -k 10 600 python -m pytest tests/test_gpu_gemm.py -q -x -k 'tn or grouped or wgrad' > gpurun_out/${TAG}_tests.log 2>&1; tail -3 gpurun_out/${TAG}_tests.log"
b() {
  local name=$1; shift
  step bench_$name bash -c "$* timeout -k 10 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --profile-steps 0 2>> gpurun_out/${TAG}_bench.err | tail -1 | python -c \"import json,sys; d=json.loads(sys.stdin.read()); print('$name', round(d['ms_per_step'],2))\" | tee -a gpurun_out/${TAG}_ab.txt"
}
for r in 1 2 3; do
b default$r
b g22_u22_$r S2T_TN_W=0 S2T_TN_GROUP_TILE=22 S2T_TN_TILE=22
b g11_u22_$r S2T_TN_W=0 S2T_TN_TILE=22
b g22_u22_b3072_$r S2T_TN_W=0 S2T_TN_GROUP_TILE=22 S2T_TN_TILE=22 S2T_TN_GROUP_BLOCKS=3072
done
