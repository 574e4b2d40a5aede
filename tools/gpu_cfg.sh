#!/bin/bash
# bench + kernel trace of one config.  usage (inside gpurun): bash tools/gpu_cfg.sh TAG CONFIG [noprof]
set -o pipefail
TAG=${1:-r03_c2}; CFG=${2:-C2}; NOPROF=${3:-}
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
python bench.py --config $CFG --steps 10 --warmup 3 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
echo "bench rc=$?"; tail -3 gpurun_out/${TAG}_bench.err; cut -c1-400 gpurun_out/${TAG}_bench.json
[ -n "$NOPROF" ] && exit 0
rm -rf gpurun_out/${TAG}_prof
rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_prof -o ${TAG} --output-format csv -- python bench.py --config $CFG --steps 5 --warmup 3 --no-cpu-baseline --profile-steps 0 > gpurun_out/${TAG}_profbench.json 2> gpurun_out/${TAG}_profbench.err
echo "prof rc=$?"
MS=$(python -c "import json;print(json.load(open('gpurun_out/${TAG}_profbench.json'))['ms_per_step'])")
python tools/prof_summary.py gpurun_out/${TAG}_prof ${TAG} 5 $MS
mkdir -p gpurun_out/profiles_${TAG} && cp profiles/${TAG}_* gpurun_out/profiles_${TAG}/
find gpurun_out/${TAG}_prof -name "*_kernel_trace.csv" -delete
head -45 profiles/${TAG}_timed_region.txt | cut -c1-200
