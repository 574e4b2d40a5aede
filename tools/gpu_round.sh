#!/bin/bash
# One GPU-box visit: GPU tests, the default bench, and a rocprofv3 kernel trace of the same bench.
# usage (inside gpurun): bash tools/gpu_round.sh TAG [pytest-args]
set -o pipefail
TAG=${1:-r02}
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
python -m pytest tests -m gpu -x -q ${@:2} > gpurun_out/${TAG}_tests.log 2>&1
echo "tests rc=$?" | tee -a gpurun_out/${TAG}_tests.log
tail -5 gpurun_out/${TAG}_tests.log
python bench.py --steps 10 --warmup 3 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
echo "bench rc=$?"; tail -3 gpurun_out/${TAG}_bench.err; cut -c1-600 gpurun_out/${TAG}_bench.json
rm -rf gpurun_out/${TAG}_prof
rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_prof -o ${TAG} --output-format csv -- python bench.py --steps 5 --warmup 3 --no-cpu-baseline --profile-steps 0 > gpurun_out/${TAG}_profbench.json 2> gpurun_out/${TAG}_profbench.err
echo "prof rc=$?"
MS=$(python -c "import json;print(json.load(open('gpurun_out/${TAG}_profbench.json'))['ms_per_step'])")
python tools/prof_summary.py gpurun_out/${TAG}_prof ${TAG} 5 $MS
mkdir -p gpurun_out/profiles_${TAG} && cp profiles/${TAG}_* gpurun_out/profiles_${TAG}/
# keep the merged-back payload small: drop the raw trace
find gpurun_out/${TAG}_prof -name "*_kernel_trace.csv" -delete
head -12 profiles/${TAG}_timed_region.txt
