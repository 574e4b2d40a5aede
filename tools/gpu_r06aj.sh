#!/bin/bash
# round 6, visit aj: which GEMM forms the plan may choose from -- in-step A/B of candidate sets
TAG=${1:-r06aj}
mkdir -p gpurun_out
export TAG
. tools/gpu_step.sh
b() {
  local name=$1; local cfg=$2; shift; shift
  step bench_$name bash -c "$* timeout -k 10 600 python bench.py --config $cfg --steps 20 --warmup 5 --no-cpu-baseline --profile-steps 0 2>> gpurun_out/${TAG}_bench.err | tail -1 | python -c \"import json,sys; d=json.loads(sys.stdin.read()); print('$name', round(d['ms_per_step'],2))\" | tee -a gpurun_out/${TAG}_ab.txt"
}
for r in 1 2; do
b all_$r C3
b noks2_$r C3 S2T_X3P_TILES2=222,321,312,411,2022,2021,2012
b dma1_$r C3 S2T_X3P_TILES2=2022,2021,2012,2011
b dma2_$r C3 S2T_X3P_TILES2=2222,2221,2212,2211
b db_$r C3 S2T_X3P_TILES2=222,321,312,411
b dma12_$r C3 S2T_X3P_TILES2=2022,2021,2012,2011,2222,2221,2212,2211
done
