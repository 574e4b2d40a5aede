#!/bin/bash
# round 6, visit b: per-class arithmetic policies -- gradient error vs the oracle per policy, and the
# C3 step time of the candidate policies
TAG=${1:-r06b}
mkdir -p gpurun_out
export TAG
. tools/gpu_step.sh
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
step gemm_tests bash -c "timeout -k 10 900 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_zip_ops.py -q -x -k 'gemm or wgrad or x3p or batched or tn_ or conv3x3' > gpurun_out/${TAG}_gemm_tests.log 2>&1; tail -5 gpurun_out/${TAG}_gemm_tests.log"
step policy bash -c "timeout -k 10 1500 python tools/exp_arith_policy.py 0 1 2 > gpurun_out/${TAG}_policy.txt 2> gpurun_out/${TAG}_policy.err; tail -30 gpurun_out/${TAG}_policy.txt; tail -3 gpurun_out/${TAG}_policy.err"
for P in 3333 2222 3222 2322 2232 3322; do
  step bench_$P bash -c "S2T_GEMM_ARITH_F=${P:0:1} S2T_GEMM_ARITH_D=${P:1:1} S2T_GEMM_ARITH_W=${P:2:1} S2T_GEMM_ARITH_S=${P:3:1} timeout -k 10 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --profile-steps 0 > gpurun_out/${TAG}_bench_$P.json 2> gpurun_out/${TAG}_bench_$P.err; tail -1 gpurun_out/${TAG}_bench_$P.json | cut -c1-200"
done
