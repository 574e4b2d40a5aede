#!/bin/bash
# round 6, visit ab: the all-waves weight-gradient form with a two-piece LDS image (32 KB) against the three-piece image (48 KB)
TAG=${1:-r06ab}
mkdir -p gpurun_out
export TAG
. tools/gpu_step.sh
step tests bash -c "timeout -k 10 900 python -m pytest tests/test_gpu_gemm.py -q -x -k 'tn or grouped or wgrad' > gpurun_out/${TAG}_tests.log 2>&1; tail -3 gpurun_out/${TAG}_tests.log"
b() {
  local name=$1; local cfg=$2; shift; shift
  step bench_$name bash -c "$* timeout -k 10 600 python bench.py --config $cfg --steps 20 --warmup 5 --no-cpu-baseline --profile-steps 0 2>> gpurun_out/${TAG}_bench.err | tail -1 | python -c \"import json,sys; d=json.loads(sys.stdin.read()); print('$name', round(d['ms_per_step'],2))\" | tee -a gpurun_out/${TAG}_ab.txt"
}
for r in 1 2 3; do
b C3_p2_$r C3
b C3_p3img_$r C3 S2T_TN_P2=0
done
b C2_p2 C2
b C2_p3img C2 S2T_TN_P2=0
