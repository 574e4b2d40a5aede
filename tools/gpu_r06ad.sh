#!/bin/bash
# round 6, visit ad: profile of the tree with the all-waves weight gradients
TAG=${1:-r06ad}
mkdir -p gpurun_out
export TAG
. tools/gpu_step.sh
step prof bash tools/gpu_prof.sh ${TAG}
