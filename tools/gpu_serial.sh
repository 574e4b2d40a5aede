#!/bin/bash
# serialized trace (every launch on the main stream) of the current tree (inside gpurun):
#   bash tools/gpu_serial.sh TAG  ->  gpurun_out/TAG_trace.csv (compact), TAG_overlap.txt
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-ser}
EXTRA="${@:2}"       # further bench.py arguments
export S2T_WGRAD_STREAM=0 S2T_CONV_W_SIDE=0 S2T_WGRAD_SIDE_MORE=0 S2T_WHITEN_STREAM=0
rm -rf gpurun_out/${TAG}_prof
rocprofv3 --kernel-trace -d gpurun_out/${TAG}_prof -o ${TAG} --output-format csv -- python bench.py --steps 5 --warmup 6 --no-cpu-baseline --profile-steps 0 $EXTRA > gpurun_out/${TAG}_profbench.json 2> gpurun_out/${TAG}_profbench.err
MS=$(python -c "import json;print(json.load(open('gpurun_out/${TAG}_profbench.json'))['ms_per_step'])")
T=$(find gpurun_out/${TAG}_prof -name "*_kernel_trace.csv")
head -1 $T > gpurun_out/${TAG}_header.txt
python tools/prof_overlap.py $T 5 $MS gpurun_out/${TAG}_trace.csv > gpurun_out/${TAG}_overlap.txt
head -3 gpurun_out/${TAG}_overlap.txt
rm -rf gpurun_out/${TAG}_prof
