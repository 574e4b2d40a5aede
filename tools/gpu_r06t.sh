#!/bin/bash
# round 6, visit t: Balancer coefficients precomputed per launch (s2t_gemm_x3p_bal), map kernels without scratch
TAG=${1:-r06t}
mkdir -p gpurun_out
export TAG
. tools/gpu_step.sh
step tests bash -c "timeout -k 10 900 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_zip_ops.py -q -x -k 'balancer or conv3x3 or dwconv or implicit' > gpurun_out/${TAG}_tests.log 2>&1; tail -4 gpurun_out/${TAG}_tests.log"
step tests2 bash -c "timeout -k 10 900 python -m pytest tests/test_gpu_zip_layer.py tests/test_gpu_full_configs.py tests/test_gpu_frontend_losses.py -q -x -k 'not c2 and not c4 and not c5' > gpurun_out/${TAG}_tests2.log 2>&1; tail -4 gpurun_out/${TAG}_tests2.log"
step fe bash -c "timeout -k 10 300 python tools/bench_frontend.py 2>&1 | tail -1 | tee gpurun_out/${TAG}_fe.txt"
b() {
  local name=$1; shift
  step bench_$name bash -c "$* timeout -k 10 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --profile-steps 0 2>> gpurun_out/${TAG}_bench.err | tail -1 | python -c \"import json,sys; d=json.loads(sys.stdin.read()); print('$name', round(d['ms_per_step'],2))\" | tee -a gpurun_out/${TAG}_ab.txt"
}
b a
b b
b c
step prof bash tools/gpu_prof.sh ${TAG}
