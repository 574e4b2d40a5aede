#!/bin/bash
# round 6, visit ag: 3x3 conv weight gradient on the all-waves form with the implicit patch operand
TAG=${1:-r06ag}
mkdir -p gpurun_out
export TAG
. tools/gpu_step.sh
step tests bash -c "timeout -k 10 900 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_zip_ops.py tests/test_gpu_conformer_layer.py -q -x -k 'conv3x3 or dwconv or implicit or subsampling or conv' > gpurun_out/${TAG}_tests.log 2>&1; tail -3 gpurun_out/${TAG}_tests.log"
step tests2 bash -c "S2T_CONV_W_P3=2 timeout -k 10 900 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_zip_ops.py tests/test_gpu_conformer_layer.py -q -x -k 'conv3x3 or dwconv or implicit or subsampling or conv' > gpurun_out/${TAG}_tests2.log 2>&1; tail -3 gpurun_out/${TAG}_tests2.log"
b() {
  local name=$1; local cfg=$2; shift; shift
  step bench_$name bash -c "$* timeout -k 10 600 python bench.py --config $cfg --steps 20 --warmup 5 --no-cpu-baseline --profile-steps 0 2>> gpurun_out/${TAG}_bench.err | tail -1 | python -c \"import json,sys; d=json.loads(sys.stdin.read()); print('$name', round(d['ms_per_step'],2))\" | tee -a gpurun_out/${TAG}_ab.txt"
}
for r in 1 2 3; do
b C3_p3_$r C3
b C3_old_$r C3 S2T_CONV_W_P3=0
done
for r in 1 2; do
b C2_W_$r C2
b C2_p3_$r C2 S2T_CONV_W_P3=2
done
