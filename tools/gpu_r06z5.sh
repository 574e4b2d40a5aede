#!/bin/bash
# round 6, second session, final visit: the whole GPU suite, the driver's bench command, kernel trace, PMC traffic,
# the other configurations and the chunked training mode
TAG=${1:-r06z5}
mkdir -p gpurun_out
export TAG
. tools/gpu_step.sh
step suite bash -c "timeout -k 10 1500 python -m pytest tests -m gpu -q -x > gpurun_out/${TAG}_suite.log 2>&1; tail -6 gpurun_out/${TAG}_suite.log"
step smoke bash -c "timeout -k 10 300 python -c 'import __graft_entry__ as g; g.smoke()' 2>&1 | tail -3"
step bench bash -c "timeout -k 10 900 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; tail -1 gpurun_out/${TAG}_bench.json | cut -c1-400"
step bench_x3 bash -c "S2T_GEMM_ARITH=3 timeout -k 10 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --profile-steps 0 2>> gpurun_out/${TAG}_bench.err | tail -1 | python -c \"import json,sys; d=json.loads(sys.stdin.read()); print('all six products', round(d['ms_per_step'],2), d['config']['gemm_arith'])\""
for cfg in C2 C4 C5; do
step bench_$cfg bash -c "timeout -k 10 600 python bench.py --config $cfg --steps 20 --warmup 5 --no-cpu-baseline --profile-steps 0 2>> gpurun_out/${TAG}_bench.err | tail -1 | python -c \"import json,sys; d=json.loads(sys.stdin.read()); print('$cfg', round(d['ms_per_step'],2))\" | tee -a gpurun_out/${TAG}_cfgs.txt"
done
step bench_chunk bash -c "timeout -k 10 600 python bench.py --random-chunk --steps 20 --warmup 5 --no-cpu-baseline --profile-steps 0 2>> gpurun_out/${TAG}_bench.err | tail -1 | python -c \"import json,sys; d=json.loads(sys.stdin.read()); print('random-chunk', round(d['ms_per_step'],2))\" | tee -a gpurun_out/${TAG}_cfgs.txt"
step bench_ddp bash -c "timeout -k 10 600 python bench.py --ddp-force allreduce --steps 20 --warmup 5 --no-cpu-baseline --profile-steps 0 2>> gpurun_out/${TAG}_bench.err | tail -1 | python -c \"import json,sys; d=json.loads(sys.stdin.read()); print('ddp forced (world 1)', round(d['ms_per_step'],2))\" | tee -a gpurun_out/${TAG}_cfgs.txt"
step prof bash tools/gpu_prof.sh ${TAG}
step pmc bash tools/gpu_pmc2.sh ${TAG} C3
