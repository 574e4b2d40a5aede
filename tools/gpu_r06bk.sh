#!/bin/bash
# round 6, visit bk: bypass_up_bwd 16-byte form, grid cap 128 / 192 / 256 alone; test; in-step A/B
TAG=${1:-r06bk}
mkdir -p gpurun_out
export TAG
. tools/gpu_step.sh
run() { echo "--- $*" >> gpurun_out/${TAG}_atomics.txt; env "$@" timeout -k 10 200 python tools/bench_atomics.py 2>&1 | grep bypass_up >> gpurun_out/${TAG}_atomics.txt; }
step a1 run S2T_BUP_BLOCKS=128
step a2 run S2T_BUP_BLOCKS=192
step a3 run S2T_BUP_BLOCKS=256
step a4 run S2T_BUP_BLOCKS=384
cat gpurun_out/${TAG}_atomics.txt
step tests bash -c "timeout -k 10 600 python -m pytest tests/test_gpu_zip_ops.py tests/test_gpu_zipformer.py -q -x > gpurun_out/${TAG}_tests.log 2>&1; tail -3 gpurun_out/${TAG}_tests.log"
b() {
  local name=$1; local cfg=$2; shift; shift
  step bench_$name bash -c "$* timeout -k 10 600 python bench.py --config $cfg --steps 40 --warmup 5 --no-cpu-baseline --profile-steps 0 2>> gpurun_out/${TAG}_bench.err | tail -1 | python -c \"import json,sys; d=json.loads(sys.stdin.read()); print('$name', round(d['ms_per_step'],2))\" | tee -a gpurun_out/${TAG}_ab.txt"
}
for r in 1 2 3; do
b new256_$r C3
b new128_$r C3 S2T_BUP_BLOCKS=128
b old_$r C3 S2T_BYPASS_UP_BWD16=0
done
