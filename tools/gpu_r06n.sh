#!/bin/bash
TAG=${1:-r06n}
mkdir -p gpurun_out
export TAG
. tools/gpu_step.sh
step tests bash -c "timeout -k 10 900 python -m pytest tests/test_gpu_zipformer.py tests/test_gpu_full_configs.py tests/test_gpu_conformer_tasks.py -q -x > gpurun_out/${TAG}_tests.log 2>&1; tail -5 gpurun_out/${TAG}_tests.log"
b() {
  local name=$1; shift
  step bench_$name bash -c "$* timeout -k 10 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --profile-steps 0 2>> gpurun_out/${TAG}_bench.err | tail -1 | python -c \"import json,sys; d=json.loads(sys.stdin.read()); print('$name', round(d['ms_per_step'],2))\" | tee -a gpurun_out/${TAG}_ab.txt"
}
b new1
b old1 S2T_ADHOC_X3P=0
b new2
b old2 S2T_ADHOC_X3P=0
b new3
b old3 S2T_ADHOC_X3P=0
