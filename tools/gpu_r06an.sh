#!/bin/bash
# round 6, visit an: kernel traces of C2 / C4 / C5 on the last tree + the frontend alone
TAG=${1:-r06an}
mkdir -p gpurun_out
export TAG
. tools/gpu_step.sh
for c in C2 C4 C5; do
  lc=$(echo $c | tr 'A-Z' 'a-z')
  step prof_$c bash -c "bash tools/gpu_prof.sh r06z2_${lc} --config $c > gpurun_out/${TAG}_${lc}.log 2>&1; head -9 profiles/r06z2_${lc}_timed_region.txt | cut -c1-160"
done
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
rm -rf gpurun_out/${TAG}_feprof
step feprof timeout -k 10 400 rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_feprof -o fe --output-format csv -- python3 tools/bench_frontend.py > gpurun_out/${TAG}_feprof.log 2>&1
MS=$(python -c "import re;t=open('gpurun_out/${TAG}_feprof.log').read();m=re.search(r'fwd ([0-9.]+) ms  bwd ([0-9.]+)',t);print(float(m.group(1))+float(m.group(2)))")
python tools/prof_summary.py gpurun_out/${TAG}_feprof r06z2_frontend 10 $MS
mkdir -p gpurun_out/profiles_${TAG} && cp profiles/r06z2_* gpurun_out/profiles_${TAG}/
find gpurun_out -name "*kernel_trace.csv" -delete
grep frontend gpurun_out/${TAG}_feprof.log
