#!/bin/bash
# round 6, visit av: workgroup targets of the two-piece-image weight-gradient kernels; coefficient-pass test
TAG=${1:-r06av}
mkdir -p gpurun_out
export TAG
. tools/gpu_step.sh
step tests bash -c "timeout -k 10 600 python -m pytest tests/test_gpu_gemm.py -q -x -k 'balancer' > gpurun_out/${TAG}_tests.log 2>&1; tail -3 gpurun_out/${TAG}_tests.log"
b() {
  local name=$1; local cfg=$2; shift; shift
  step bench_$name bash -c "$* timeout -k 10 600 python bench.py --config $cfg --steps 20 --warmup 5 --no-cpu-baseline --profile-steps 0 2>> gpurun_out/${TAG}_bench.err | tail -1 | python -c \"import json,sys; d=json.loads(sys.stdin.read()); print('$name', round(d['ms_per_step'],2))\" | tee -a gpurun_out/${TAG}_ab.txt"
}
for r in 1 2; do
b base_$r C3
b g1024_$r C3 S2T_TN_GROUP_BLOCKS=1024
b g2048_$r C3 S2T_TN_GROUP_BLOCKS=2048
b g3072_$r C3 S2T_TN_GROUP_BLOCKS=3072
b u1536_$r C3 S2T_TN_BLOCKS=1536
b u512_$r C3 S2T_TN_BLOCKS=512
done
