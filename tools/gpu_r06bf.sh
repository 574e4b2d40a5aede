#!/bin/bash
# round 6, visit bf: S2T_SIDE_DEFER bits, same-box A/B (40 timed steps each, 3 rounds)
TAG=${1:-r06bf}
mkdir -p gpurun_out
export TAG
. tools/gpu_step.sh
b() {
  local name=$1; local cfg=$2; shift; shift
  step bench_$name bash -c "$* timeout -k 10 600 python bench.py --config $cfg --steps 40 --warmup 5 --no-cpu-baseline --profile-steps 0 2>> gpurun_out/${TAG}_bench.err | tail -1 | python -c \"import json,sys; d=json.loads(sys.stdin.read()); print('$name', round(d['ms_per_step'],2))\" | tee -a gpurun_out/${TAG}_ab.txt"
}
for r in 1 2 3; do
b d0_$r C3 S2T_SIDE_DEFER=0
b d1_$r C3 S2T_SIDE_DEFER=1
b d3_$r C3 S2T_SIDE_DEFER=3
b d7_$r C3 S2T_SIDE_DEFER=7
done
