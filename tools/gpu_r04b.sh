#!/bin/bash
# round-4 measurement visit: rocprof of the bench, host profile, DP overhead, accum 20, other configs
set -o pipefail
TAG=${1:-r04b}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() { python bench.py --steps 10 --warmup 6 --no-cpu-baseline --profile-steps 0 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', '->', round(d['ms_per_step'],2), 'ms/step', round(d['value']), 'audio-s/s')"; }
run
run --ddp-force allreduce
run --ddp-force rs_ag
run
run --accum 20 --steps 20
run --random-chunk
run --config C2
run --config C4
run --config C5
python tools/exp_host.py > gpurun_out/${TAG}_host.log 2>&1; head -3 gpurun_out/${TAG}_host.log | grep -v amdgpu
rm -rf gpurun_out/${TAG}_prof
rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_prof -o ${TAG} --output-format csv -- python bench.py --steps 5 --warmup 6 --no-cpu-baseline --profile-steps 0 > gpurun_out/${TAG}_profbench.json 2> gpurun_out/${TAG}_profbench.err
MS=$(python -c "import json;print(json.load(open('gpurun_out/${TAG}_profbench.json'))['ms_per_step'])")
python tools/prof_summary.py gpurun_out/${TAG}_prof ${TAG} 5 $MS
mkdir -p gpurun_out/profiles_${TAG} && cp profiles/${TAG}_* gpurun_out/profiles_${TAG}/
find gpurun_out/${TAG}_prof -name "*_kernel_trace.csv" -delete
head -12 profiles/${TAG}_timed_region.txt
