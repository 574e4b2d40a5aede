#!/bin/bash
# round 6, visit ai: the 8-wave workgroups of the LDS-DMA GEMM (tile codes 3000 +): parity, isolated table, step A/B
TAG=${1:-r06ai}
mkdir -p gpurun_out
export TAG
. tools/gpu_step.sh
step tests bash -c "timeout -k 10 600 python -m pytest tests/test_gpu_gemm.py -q -x -k 'x3p' > gpurun_out/${TAG}_tests.log 2>&1; tail -3 gpurun_out/${TAG}_tests.log"
step table bash -c "X3P_CFGS=0,222,321,312,411,322,2022,2021,2012,2011,2222,2221,2212,2211,3022,3021,3012,3222,3212 timeout -k 10 600 python tools/bench_x3p.py > gpurun_out/${TAG}_x3p_table.txt 2>&1; grep -E 'sum lt' gpurun_out/${TAG}_x3p_table.txt"
b() {
  local name=$1; local cfg=$2; shift; shift
  step bench_$name bash -c "$* timeout -k 10 600 python bench.py --config $cfg --steps 20 --warmup 5 --no-cpu-baseline --profile-steps 0 2>> gpurun_out/${TAG}_bench.err | tail -1 | python -c \"import json,sys; d=json.loads(sys.stdin.read()); print('$name', round(d['ms_per_step'],2))\" | tee -a gpurun_out/${TAG}_ab.txt"
}
T8=222,321,312,411,2022,2021,2012,2222,2221,2212,2211,3022,3021,3012,3222,3212
for r in 1 2 3; do
b base_$r C3
b w8_$r C3 S2T_X3P_TILES2=$T8
done
