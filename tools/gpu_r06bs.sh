#!/bin/bash
# round 6, visit bs: the frontend's output Linear as its own node (weight gradient on the side stream) -- tests + A/B
TAG=${1:-r06bs}
mkdir -p gpurun_out
export TAG
. tools/gpu_step.sh
step tests bash -c "timeout -k 10 900 python -m pytest tests/test_gpu_zipformer.py tests/test_gpu_full_configs.py tests/test_gpu_ddp.py -q -x > gpurun_out/${TAG}_tests.log 2>&1; tail -3 gpurun_out/${TAG}_tests.log"
b() {
  local name=$1; local cfg=$2; shift; shift
  step bench_$name bash -c "$* timeout -k 10 600 python bench.py --config $cfg --steps 40 --warmup 5 --no-cpu-baseline --profile-steps 0 2>> gpurun_out/${TAG}_bench.err | tail -1 | python -c \"import json,sys; d=json.loads(sys.stdin.read()); print('$name', round(d['ms_per_step'],2))\" | tee -a gpurun_out/${TAG}_ab.txt"
}
for r in 1 2 3; do
b side_$r C3
b main_$r C3 S2T_FRONT_W_SIDE=3
done
