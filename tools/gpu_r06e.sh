#!/bin/bash
# round 6, visit e: 16x16x32 product form (kernel tests + isolated timings), the suite with every
# plan decision forced to the own two-piece kernels (which tests hold fp32-level bounds?), step A/B
TAG=${1:-r06e}
mkdir -p gpurun_out
export TAG
. tools/gpu_step.sh
step gemm_tests bash -c "timeout -k 10 900 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_heads.py tests/test_gpu_zip_ops.py -q -x > gpurun_out/${TAG}_gemm_tests.log 2>&1; tail -6 gpurun_out/${TAG}_gemm_tests.log"
step x3p bash -c "timeout -k 10 600 python tools/bench_x3p.py > gpurun_out/${TAG}_x3p.txt 2>&1; tail -12 gpurun_out/${TAG}_x3p.txt"
step suite_forced bash -c "S2T_X3P_MARGIN=100 S2T_LT_OWN_MARGIN=100 timeout -k 10 1500 python -m pytest tests -m gpu -q > gpurun_out/${TAG}_suite_forced.log 2>&1; tail -25 gpurun_out/${TAG}_suite_forced.log"
for i in 1 2; do
  for V in with16 without16; do
    E=""; [ $V = without16 ] && E="S2T_X3P_TILES2=222,321,312,411,2022,2021,2012,2222,2221,2212,2211"
    step bench_$V bash -c "$E timeout -k 10 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --profile-steps 0 2>> gpurun_out/${TAG}_bench.err | tail -1 | python -c \"import json,sys; d=json.loads(sys.stdin.read()); print('$V', round(d['ms_per_step'],2))\" | tee -a gpurun_out/${TAG}_ab.txt"
  done
done
