#!/bin/bash
# round 6, visit a: the two-piece arithmetic -- kernel tests in both modes, isolated GEMM timings,
# the C3 step in both modes, then the whole GPU suite under S2T_GEMM_ARITH=2 (which parity tests hold?)
TAG=${1:-r06a}
mkdir -p gpurun_out
export TAG
. tools/gpu_step.sh
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
step gemm_tests bash -c "timeout -k 10 900 python -m pytest tests/test_gpu_gemm.py -q -x > gpurun_out/${TAG}_gemm_tests.log 2>&1; tail -15 gpurun_out/${TAG}_gemm_tests.log"
for A in 3 2; do
  step x3p_a$A bash -c "S2T_GEMM_ARITH=$A timeout -k 10 600 python tools/bench_x3p.py > gpurun_out/${TAG}_x3p_a$A.txt 2>&1; tail -12 gpurun_out/${TAG}_x3p_a$A.txt"
  step tn_a$A bash -c "S2T_GEMM_ARITH=$A timeout -k 10 300 python tools/bench_tn.py > gpurun_out/${TAG}_tn_a$A.txt 2>&1; tail -3 gpurun_out/${TAG}_tn_a$A.txt"
done
for A in 3 2 3 2; do
  step bench_a$A bash -c "S2T_GEMM_ARITH=$A timeout -k 10 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --profile-steps 0 >> gpurun_out/${TAG}_bench_a$A.json 2>> gpurun_out/${TAG}_bench_a$A.err; tail -1 gpurun_out/${TAG}_bench_a$A.json | cut -c1-400"
done
step suite_a2 bash -c "S2T_GEMM_ARITH=2 timeout -k 10 1500 python -m pytest tests -m gpu -q > gpurun_out/${TAG}_suite_a2.log 2>&1; tail -40 gpurun_out/${TAG}_suite_a2.log"
step new_tests_a3 bash -c "S2T_GEMM_ARITH=3 timeout -k 10 900 python -m pytest tests/test_gpu_full_configs.py tests/test_gpu_zip_layer.py tests/test_gpu_validation.py tests/test_gpu_bench.py -q > gpurun_out/${TAG}_new_a3.log 2>&1; tail -25 gpurun_out/${TAG}_new_a3.log"
