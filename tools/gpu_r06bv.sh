#!/bin/bash
# round 6, visit bv: does the default warm-up (3 steps) leave the timed steps cold?
TAG=${1:-r06bv}
mkdir -p gpurun_out
export TAG
. tools/gpu_step.sh
b() {
  local name=$1; shift
  step bench_$name bash -c "timeout -k 10 600 python bench.py --no-cpu-baseline --profile-steps 0 $* 2>> gpurun_out/${TAG}_bench.err | tail -1 | python -c \"import json,sys; d=json.loads(sys.stdin.read()); print('$name', round(d['ms_per_step'],2))\" | tee -a gpurun_out/${TAG}_ab.txt"
}
for r in 1 2; do
b w3s10_$r --steps 10 --warmup 3
b w6s10_$r --steps 10 --warmup 6
b w10s10_$r --steps 10 --warmup 10
b w5s40_$r --steps 40 --warmup 5
done
