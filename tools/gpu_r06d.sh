#!/bin/bash
# round 6, visit d: Balancer 16-byte form (tests + same-box A/B), simple-loss products on own kernels,
# profile of the default step
TAG=${1:-r06d}
mkdir -p gpurun_out
export TAG
. tools/gpu_step.sh
step tests bash -c "timeout -k 10 900 python -m pytest tests/test_gpu_zip_ops.py tests/test_gpu_frontend_losses.py tests/test_gpu_heads.py tests/test_gpu_zip_layer.py -q -x > gpurun_out/${TAG}_tests.log 2>&1; tail -8 gpurun_out/${TAG}_tests.log"
for i in 1 2; do
  for V in 1 0; do
    step bench_vec$V bash -c "S2T_BAL_VEC=$V timeout -k 10 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --profile-steps 0 2>> gpurun_out/${TAG}_bench.err | tail -1 | python -c \"import json,sys; d=json.loads(sys.stdin.read()); print('BAL_VEC=$V', round(d['ms_per_step'],2))\" | tee -a gpurun_out/${TAG}_ab.txt"
  done
done
step prof bash tools/gpu_prof.sh ${TAG}
