#!/bin/bash
# rocprofv3 kernel trace of a short bench run -> profiles/TAG_* summaries (also under gpurun_out/)
# usage (inside gpurun): [ENV=..] bash tools/gpu_prof.sh TAG [bench args]
set -o pipefail
TAG=${1:-prof}; shift
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
rm -rf gpurun_out/${TAG}_prof
rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_prof -o ${TAG} --output-format csv -- python bench.py --steps 5 --warmup 3 --no-cpu-baseline --profile-steps 0 "$@" > gpurun_out/${TAG}_profbench.json 2> gpurun_out/${TAG}_profbench.err
echo "prof rc=$?"
MS=$(python -c "import json;print(json.load(open('gpurun_out/${TAG}_profbench.json'))['ms_per_step'])")
python tools/prof_summary.py gpurun_out/${TAG}_prof ${TAG} 5 $MS
mkdir -p gpurun_out/profiles_${TAG} && cp profiles/${TAG}_* gpurun_out/profiles_${TAG}/
find gpurun_out/${TAG}_prof -name "*_kernel_trace.csv" -delete
head -40 profiles/${TAG}_timed_region.txt | cut -c1-150
