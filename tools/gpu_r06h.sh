#!/bin/bash
# round 6, visit h: whiten penalty product on DMA tiles / two pieces -- gradient tests + A/B + suite
TAG=${1:-r06h}
mkdir -p gpurun_out
export TAG
. tools/gpu_step.sh
step grad_tests bash -c "timeout -k 10 900 python -m pytest tests/test_gpu_full_configs.py tests/test_gpu_zip_ops.py tests/test_gpu_zip_layer.py -q -x -k 'c3 or whiten or executor' > gpurun_out/${TAG}_grad_tests.log 2>&1; tail -5 gpurun_out/${TAG}_grad_tests.log"
for i in 1 2; do
  for V in 2 0; do
    step bench_wh$V bash -c "S2T_WHITEN_X3P=$V timeout -k 10 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --profile-steps 0 2>> gpurun_out/${TAG}_bench.err | tail -1 | python -c \"import json,sys; d=json.loads(sys.stdin.read()); print('WHITEN_X3P=$V', round(d['ms_per_step'],2))\" | tee -a gpurun_out/${TAG}_ab.txt"
  done
done
step suite bash -c "timeout -k 10 1500 python -m pytest tests -m gpu -q > gpurun_out/${TAG}_suite.log 2>&1; tail -6 gpurun_out/${TAG}_suite.log"
