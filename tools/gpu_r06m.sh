#!/bin/bash
# round 6, visit m: C2 / C4 / C5 -- step time under the built-in policy and all-six-products, kernel traces
TAG=${1:-r06m}
mkdir -p gpurun_out
export TAG
. tools/gpu_step.sh
b() {
  local name=$1; shift
  step bench_$name bash -c "$* 2>> gpurun_out/${TAG}_bench.err | tail -1 | python -c \"import json,sys; d=json.loads(sys.stdin.read()); print('$name', round(d['ms_per_step'],2), d['config']['gemm_arith'], d['config']['gemm_paths'])\" | tee -a gpurun_out/${TAG}_ab.txt"
}
for C in C2 C4 C5; do
  b ${C}_policy timeout -k 10 600 python bench.py --config $C --steps 20 --warmup 5 --no-cpu-baseline --profile-steps 0
  b ${C}_x3 S2T_GEMM_ARITH=3 timeout -k 10 600 python bench.py --config $C --steps 20 --warmup 5 --no-cpu-baseline --profile-steps 0
done
for C in C2 C4 C5; do
  lc=$(echo $C | tr 'C' 'c')
  step prof_$C bash tools/gpu_prof.sh r06_${lc} --config $C
done
