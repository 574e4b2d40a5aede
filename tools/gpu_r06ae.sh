#!/bin/bash
# round 6, visit ae: wave priority of the main stream's GEMM kernels against the side stream's (s_setprio)
TAG=${1:-r06ae}
mkdir -p gpurun_out
export TAG
. tools/gpu_step.sh
b() {
  local name=$1; local cfg=$2; shift; shift
  step bench_$name bash -c "$* timeout -k 10 600 python bench.py --config $cfg --steps 20 --warmup 5 --no-cpu-baseline --profile-steps 0 2>> gpurun_out/${TAG}_bench.err | tail -1 | python -c \"import json,sys; d=json.loads(sys.stdin.read()); print('$name', round(d['ms_per_step'],2))\" | tee -a gpurun_out/${TAG}_ab.txt"
}
for r in 1 2 3; do
b prio1_$r C3
b prio2_$r C3 S2T_X3P_PRIO=2
b prio0_$r C3 S2T_X3P_PRIO=0
done
