#!/bin/bash
# round 6, visit s: the frontend alone (tools/bench_frontend.py) with its kernel trace
TAG=${1:-r06s}
mkdir -p gpurun_out
export TAG
. tools/gpu_step.sh
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
step fe bash -c "timeout -k 10 300 python tools/bench_frontend.py 2>&1 | tail -1 | tee gpurun_out/${TAG}_fe.txt"
rm -rf gpurun_out/${TAG}_feprof
step feprof timeout -k 10 400 rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_feprof -o fe --output-format csv -- python3 tools/bench_frontend.py > gpurun_out/${TAG}_feprof.log 2>&1
MS=$(python -c "import re;t=open('gpurun_out/${TAG}_feprof.log').read();m=re.search(r'fwd ([0-9.]+) ms  bwd ([0-9.]+)',t);print(float(m.group(1))+float(m.group(2)))")
python tools/prof_summary.py gpurun_out/${TAG}_feprof ${TAG}_frontend 10 $MS
mkdir -p gpurun_out/profiles_${TAG} && cp profiles/${TAG}_frontend_* gpurun_out/profiles_${TAG}/
find gpurun_out/${TAG}_feprof -name "*kernel_trace.csv" -delete
head -60 profiles/${TAG}_frontend_timed_region.txt | cut -c1-190
