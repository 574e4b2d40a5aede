#!/bin/bash
# round 6, visit i: run-to-run spread of the plan (same tree, 4 runs), tunables A/B, ATen sites, stack cost
TAG=${1:-r06i}
mkdir -p gpurun_out
export TAG
. tools/gpu_step.sh
b() {  # name, env...
  local name=$1; shift
  step bench_$name bash -c "$* timeout -k 10 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --profile-steps 0 2>> gpurun_out/${TAG}_bench.err | tail -1 | python -c \"import json,sys; d=json.loads(sys.stdin.read()); print('$name', round(d['ms_per_step'],2))\" | tee -a gpurun_out/${TAG}_ab.txt"
}
b base1
b base2
b tnw384 S2T_TN_W_BLOCKS=384
b tnw768 S2T_TN_W_BLOCKS=768
b margin90 S2T_X3P_MARGIN=0.90
b margin105 S2T_X3P_MARGIN=1.05
b base3
b nodma S2T_X3P_TILES2=222,321,312,411,322
b onlydma S2T_X3P_TILES2=2022,2021,2012,2011,2222,2221,2212,2211
b base4
step aten bash -c "timeout -k 10 600 python tools/exp_aten_sites.py > gpurun_out/${TAG}_aten.txt 2>&1; head -45 gpurun_out/${TAG}_aten.txt"
step stack bash -c "timeout -k 10 600 python tools/exp_stack_cost.py > gpurun_out/${TAG}_stack.txt 2>&1; tail -20 gpurun_out/${TAG}_stack.txt"
