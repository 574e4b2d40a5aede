#!/bin/bash
# main-stream kernels with / without the side stream (inside gpurun): bash tools/gpu_overlap.sh
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for V in 1 0; do
  export S2T_WGRAD_STREAM=$V S2T_CONV_W_SIDE=$V S2T_WGRAD_SIDE_MORE=$V
  python bench.py --steps 10 --warmup 6 --no-cpu-baseline > gpurun_out/ov${V}_bench.json 2> gpurun_out/ov${V}_bench.err
  python -c "import json;d=json.load(open('gpurun_out/ov${V}_bench.json'));print('side=$V unprofiled', round(d['ms_per_step'],2),'ms/step')"
  rm -rf gpurun_out/ov${V}_prof
  rocprofv3 --kernel-trace -d gpurun_out/ov${V}_prof -o ov${V} --output-format csv -- python bench.py --steps 5 --warmup 6 --no-cpu-baseline --profile-steps 0 > gpurun_out/ov${V}_profbench.json 2> gpurun_out/ov${V}_profbench.err
  MS=$(python -c "import json;print(json.load(open('gpurun_out/ov${V}_profbench.json'))['ms_per_step'])")
  T=$(find gpurun_out/ov${V}_prof -name "*_kernel_trace.csv")
  python tools/prof_overlap.py $T 5 $MS gpurun_out/ov${V}_trace.csv > gpurun_out/ov${V}_overlap.txt
  head -3 gpurun_out/ov${V}_overlap.txt
  rm -rf gpurun_out/ov${V}_prof
done
