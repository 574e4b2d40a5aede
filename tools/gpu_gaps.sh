#!/bin/bash
# compact kernel trace of 3 steps + main-stream gap list (inside gpurun): bash tools/gpu_gaps.sh TAG
set -o pipefail
TAG=${1:-gaps}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; rm -rf gpurun_out/${TAG}_prof
rocprofv3 --kernel-trace -d gpurun_out/${TAG}_prof -o ${TAG} --output-format csv -- python bench.py --steps 3 --warmup 4 --no-cpu-baseline --profile-steps 0 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
MS=$(python -c "import json;print(json.load(open('gpurun_out/${TAG}_bench.json'))['ms_per_step'])")
T=$(find gpurun_out/${TAG}_prof -name "*_kernel_trace.csv")
python tools/prof_overlap.py $T 3 $MS gpurun_out/${TAG}_trace.csv > gpurun_out/${TAG}_overlap.txt
python tools/prof_gaps.py gpurun_out/${TAG}_trace.csv 40 > gpurun_out/${TAG}_gaps.txt
rm -rf gpurun_out/${TAG}_prof
tail -5 gpurun_out/${TAG}_gaps.txt
