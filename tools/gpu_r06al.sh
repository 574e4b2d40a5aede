#!/bin/bash
# round 6, visit al: 32-deep barrier intervals for the implicit-operand GEMM (conv forward / gather data gradient)
TAG=${1:-r06al}
mkdir -p gpurun_out
export TAG
. tools/gpu_step.sh
step tests bash -c "S2T_CONV_MAP_TILE=222 S2T_CONV_MAP_DGRAD_TILE=221 timeout -k 10 600 python -m pytest tests/test_gpu_zip_ops.py tests/test_gpu_gemm.py tests/test_gpu_conformer_layer.py -q -x -k 'conv3x3 or dwconv or implicit or subsampling or conv' > gpurun_out/${TAG}_tests.log 2>&1; tail -3 gpurun_out/${TAG}_tests.log"
step fe0 bash -c "timeout -k 10 300 python tools/bench_frontend.py 2>&1 | tail -1"
step fe1 bash -c "S2T_CONV_MAP_TILE=222 S2T_CONV_MAP_DGRAD_TILE=221 timeout -k 10 300 python tools/bench_frontend.py 2>&1 | tail -1"
step fe2 bash -c "S2T_CONV_MAP_TILE=221 S2T_CONV_MAP_DGRAD_TILE=211 timeout -k 10 300 python tools/bench_frontend.py 2>&1 | tail -1"
b() {
  local name=$1; local cfg=$2; shift; shift
  step bench_$name bash -c "$* timeout -k 10 600 python bench.py --config $cfg --steps 20 --warmup 5 --no-cpu-baseline --profile-steps 0 2>> gpurun_out/${TAG}_bench.err | tail -1 | python -c \"import json,sys; d=json.loads(sys.stdin.read()); print('$name', round(d['ms_per_step'],2))\" | tee -a gpurun_out/${TAG}_ab.txt"
}
for r in 1 2; do
b ks1_$r C3
b ks2_$r C3 S2T_CONV_MAP_TILE=222 S2T_CONV_MAP_DGRAD_TILE=221
b ks2b_$r C3 S2T_CONV_MAP_TILE=221 S2T_CONV_MAP_DGRAD_TILE=211
done
b C2_ks1 C2
b C2_ks2 C2 S2T_CONV_MAP_TILE=222
