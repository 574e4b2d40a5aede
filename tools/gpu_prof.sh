#!/bin/bash
# rocprofv3 kernel trace of the default bench (5 timed steps) -> profiles/${TAG}_{timed_region.txt,kernel_stats.csv}
# usage (inside a gpurun visit, after `. tools/gpu_step.sh`): bash tools/gpu_prof.sh TAG [bench args]
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
rm -rf gpurun_out/${TAG}_prof
rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_prof -o ${TAG} --output-format csv -- python bench.py --steps 5 --warmup 3 --no-cpu-baseline --profile-steps 0 "$@" > gpurun_out/${TAG}_profbench.json 2> gpurun_out/${TAG}_profbench.err
rc=$?
[ $rc -ne 0 ] && { tail -5 gpurun_out/${TAG}_profbench.err; exit $rc; }
MS=$(python -c "import json;print([json.loads(l) for l in open('gpurun_out/${TAG}_profbench.json') if l.startswith('{')][-1]['ms_per_step'])")
python tools/prof_summary.py gpurun_out/${TAG}_prof ${TAG} 5 $MS
mkdir -p gpurun_out/profiles_${TAG} && cp profiles/${TAG}_* gpurun_out/profiles_${TAG}/
find gpurun_out/${TAG}_prof -name "*_kernel_trace.csv" -delete
head -70 profiles/${TAG}_timed_region.txt | cut -c1-170
