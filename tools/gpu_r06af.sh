#!/bin/bash
# round 6, visit af: the whole GPU suite + smoke + the driver's bench on the tree with the all-waves weight gradients
TAG=${1:-r06af}
mkdir -p gpurun_out
export TAG
. tools/gpu_step.sh
step suite bash -c "timeout -k 10 1100 python -m pytest tests -m gpu -q -x > gpurun_out/${TAG}_suite.log 2>&1; tail -6 gpurun_out/${TAG}_suite.log"
step smoke bash -c "timeout -k 10 300 python -c 'import __graft_entry__ as g; g.smoke()' 2>&1 | tail -3"
step bench bash -c "timeout -k 10 900 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; tail -1 gpurun_out/${TAG}_bench.json | cut -c1-300"
