#!/bin/bash
# round 6, visit bm: 16-byte norm_bypass_bwd -- test, alone at caps, in-step A/B
TAG=${1:-r06bm}
mkdir -p gpurun_out
export TAG
. tools/gpu_step.sh
step tests bash -c "timeout -k 10 600 python -m pytest tests/test_gpu_zip_ops.py tests/test_gpu_zip_layer.py -q -x > gpurun_out/${TAG}_tests.log 2>&1; tail -3 gpurun_out/${TAG}_tests.log"
run() { echo "--- $*" >> gpurun_out/${TAG}_atomics.txt; env "$@" timeout -k 10 200 python tools/bench_atomics.py 2>&1 | grep norm_bypass >> gpurun_out/${TAG}_atomics.txt; }
step a0 run S2T_NB_BWD16=0
step a1 run S2T_NB_BWD_BLOCKS=256
step a2 run S2T_NB_BWD_BLOCKS=512
step a3 run S2T_NB_BWD_BLOCKS=1024
cat gpurun_out/${TAG}_atomics.txt
b() {
  local name=$1; local cfg=$2; shift; shift
  step bench_$name bash -c "$* timeout -k 10 600 python bench.py --config $cfg --steps 40 --warmup 5 --no-cpu-baseline --profile-steps 0 2>> gpurun_out/${TAG}_bench.err | tail -1 | python -c \"import json,sys; d=json.loads(sys.stdin.read()); print('$name', round(d['ms_per_step'],2))\" | tee -a gpurun_out/${TAG}_ab.txt"
}
for r in 1 2 3; do
b new512_$r C3
b new256_$r C3 S2T_NB_BWD_BLOCKS=256
b old_$r C3 S2T_NB_BWD16=0
done
