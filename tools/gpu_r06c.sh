#!/bin/bash
# round 6, visit c: the built-in policy (F2 D2 W2 S3): penalty-product class experiment, the whole
# GPU suite, bench + rocprof trace
TAG=${1:-r06c}
mkdir -p gpurun_out
export TAG
. tools/gpu_step.sh
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
step policy_pgD bash -c "S2T_WHITEN_PG_CLS=1 POLICIES=2223 timeout -k 10 900 python tools/exp_arith_policy.py 0 1 > gpurun_out/${TAG}_policy_pgD.txt 2> gpurun_out/${TAG}_policy_pgD.err; cut -c1-330 gpurun_out/${TAG}_policy_pgD.txt"
step suite bash -c "timeout -k 10 1500 python -m pytest tests -m gpu -q > gpurun_out/${TAG}_suite.log 2>&1; tail -15 gpurun_out/${TAG}_suite.log"
for P in default pgD; do
  E=""; [ $P = pgD ] && E="S2T_WHITEN_PG_CLS=1"
  step bench_$P bash -c "$E timeout -k 10 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --profile-steps 0 > gpurun_out/${TAG}_bench_$P.json 2> gpurun_out/${TAG}_bench_$P.err; tail -1 gpurun_out/${TAG}_bench_$P.json | cut -c1-200"
done
step bench_full bash -c "timeout -k 10 900 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; tail -1 gpurun_out/${TAG}_bench.json | cut -c1-300"
rm -rf gpurun_out/${TAG}_prof
step prof rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_prof -o ${TAG} --output-format csv -- python bench.py --steps 5 --warmup 3 --no-cpu-baseline --profile-steps 0 > gpurun_out/${TAG}_profbench.json 2> gpurun_out/${TAG}_profbench.err
MS=$(python -c "import json;print(json.load(open('gpurun_out/${TAG}_profbench.json'))['ms_per_step'])")
python tools/prof_summary.py gpurun_out/${TAG}_prof ${TAG} 5 $MS
mkdir -p gpurun_out/profiles_${TAG} && cp profiles/${TAG}_* gpurun_out/profiles_${TAG}/
find gpurun_out/${TAG}_prof -name "*_kernel_trace.csv" -delete
head -60 profiles/${TAG}_timed_region.txt
