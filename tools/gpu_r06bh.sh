#!/bin/bash
# round 6, visit bh: 16-byte form of the upsample + bypass backward (S2T_BYPASS_UP_BWD16) -- test + alone + in-step
TAG=${1:-r06bh}
mkdir -p gpurun_out
export TAG
. tools/gpu_step.sh
step tests bash -c "timeout -k 10 600 python -m pytest tests/test_gpu_zip_ops.py -q -x -k 'downsample or upsample or bypass' > gpurun_out/${TAG}_tests.log 2>&1; tail -3 gpurun_out/${TAG}_tests.log"
step elem_new bash -c "timeout -k 10 300 python tools/bench_elem.py 2>&1 | grep -i 'bypass_up' | tee gpurun_out/${TAG}_elem_new.txt"
step elem_old bash -c "S2T_BYPASS_UP_BWD16=0 timeout -k 10 300 python tools/bench_elem.py 2>&1 | grep -i 'bypass_up' | tee gpurun_out/${TAG}_elem_old.txt"
b() {
  local name=$1; local cfg=$2; shift; shift
  step bench_$name bash -c "$* timeout -k 10 600 python bench.py --config $cfg --steps 40 --warmup 5 --no-cpu-baseline --profile-steps 0 2>> gpurun_out/${TAG}_bench.err | tail -1 | python -c \"import json,sys; d=json.loads(sys.stdin.read()); print('$name', round(d['ms_per_step'],2))\" | tee -a gpurun_out/${TAG}_ab.txt"
}
for r in 1 2 3; do
b new_$r C3
b old_$r C3 S2T_BYPASS_UP_BWD16=0
done
