#!/bin/bash
# round 6, visit q: the slab attention backward at 8 waves (2 per SIMD, spills) against 4 waves and the pair
TAG=${1:-r06q}
mkdir -p gpurun_out
export TAG
. tools/gpu_step.sh
step tests bash -c "S2T_ATTN_BWD_SLAB=8 timeout -k 10 600 python -m pytest tests/test_gpu_zip_ops.py -q -x -k 'relpos or attn' > gpurun_out/${TAG}_tests.log 2>&1; tail -8 gpurun_out/${TAG}_tests.log"
b() {
  local name=$1; shift
  step bench_$name bash -c "$* timeout -k 10 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --profile-steps 0 2>> gpurun_out/${TAG}_bench.err | tail -1 | python -c \"import json,sys; d=json.loads(sys.stdin.read()); print('$name', round(d['ms_per_step'],2))\" | tee -a gpurun_out/${TAG}_ab.txt"
}
b slab8a S2T_ATTN_BWD_SLAB=8
b pair1 S2T_ATTN_BWD_SLAB=0
b slab8b S2T_ATTN_BWD_SLAB=8
b pair2 S2T_ATTN_BWD_SLAB=0
