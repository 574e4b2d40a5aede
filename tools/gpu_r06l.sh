#!/bin/bash
TAG=${1:-r06l}
mkdir -p gpurun_out
export TAG
. tools/gpu_step.sh
step tests bash -c "timeout -k 10 900 python -m pytest tests/test_gpu_zip_ops.py tests/test_gpu_zipformer.py -q -x -k 'balancer or zipformer or subsampling or frontend' > gpurun_out/${TAG}_tests.log 2>&1; tail -5 gpurun_out/${TAG}_tests.log"
b() {
  local name=$1; shift
  step bench_$name bash -c "$* timeout -k 10 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --profile-steps 0 2>> gpurun_out/${TAG}_bench.err | tail -1 | python -c \"import json,sys; d=json.loads(sys.stdin.read()); print('$name', round(d['ms_per_step'],2))\" | tee -a gpurun_out/${TAG}_ab.txt"
}
b small1
b old1 S2T_BAL_SMALL=0
b small2
b old2 S2T_BAL_SMALL=0
b small3
b old3 S2T_BAL_SMALL=0
