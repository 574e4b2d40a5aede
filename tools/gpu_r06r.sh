#!/bin/bash
# round 6, visit r: gather-form data gradient of the frontend's 32 -> 128 stride-(1,2) convolution
TAG=${1:-r06r}
mkdir -p gpurun_out
export TAG
. tools/gpu_step.sh
step tests bash -c "timeout -k 10 600 python -m pytest tests/test_gpu_zip_ops.py -q -x -k 'conv3x3 or dwconv' > gpurun_out/${TAG}_tests.log 2>&1; tail -8 gpurun_out/${TAG}_tests.log"
b() {
  local name=$1; shift
  step bench_$name bash -c "$* timeout -k 10 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --profile-steps 0 2>> gpurun_out/${TAG}_bench.err | tail -1 | python -c \"import json,sys; d=json.loads(sys.stdin.read()); print('$name', round(d['ms_per_step'],2))\" | tee -a gpurun_out/${TAG}_ab.txt"
}
b map21a
b col2im1 S2T_CONV_MAP_DGRAD=0
b map11a S2T_CONV_MAP_DGRAD_TILE=11
b map21b
b col2im2 S2T_CONV_MAP_DGRAD=0
b map11b S2T_CONV_MAP_DGRAD_TILE=11
step prof bash tools/gpu_prof.sh ${TAG}
