#!/bin/bash
# round 6, visit p: the one-pass attention backward (slab kernel) -- parity, isolated time, step A/B
TAG=${1:-r06p}
mkdir -p gpurun_out
export TAG
. tools/gpu_step.sh
step tests bash -c "timeout -k 10 600 python -m pytest tests/test_gpu_zip_ops.py -q -x -k 'relpos or attn' > gpurun_out/${TAG}_tests.log 2>&1; tail -8 gpurun_out/${TAG}_tests.log"
step tests2 bash -c "timeout -k 10 900 python -m pytest tests/test_gpu_zipformer.py tests/test_gpu_zip_layer.py tests/test_gpu_full_configs.py -q -x -k 'not c2 and not c4 and not c5' > gpurun_out/${TAG}_tests2.log 2>&1; tail -6 gpurun_out/${TAG}_tests2.log"
b() {
  local name=$1; shift
  step bench_$name bash -c "$* timeout -k 10 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --profile-steps 0 2>> gpurun_out/${TAG}_bench.err | tail -1 | python -c \"import json,sys; d=json.loads(sys.stdin.read()); print('$name', round(d['ms_per_step'],2))\" | tee -a gpurun_out/${TAG}_ab.txt"
}
b slab1
b pair1 S2T_ATTN_BWD_SLAB=0
b slab2
b pair2 S2T_ATTN_BWD_SLAB=0
b slab3
b pair3 S2T_ATTN_BWD_SLAB=0
step prof bash tools/gpu_prof.sh ${TAG}
