#!/bin/bash
# round 6, visit bc: which part of S2T_SIDE_DEFER breaks the chunked YAML-dims gradient test
TAG=${1:-r06bc}
mkdir -p gpurun_out
export TAG
. tools/gpu_step.sh
for bits in ${BITS:-2 7}; do
step tests_$bits bash -c "S2T_SIDE_DEFER=$bits timeout -k 10 900 python -m pytest tests/test_gpu_full_configs.py -q -k 'c3_yaml_dims_training_step' > gpurun_out/${TAG}_tests_$bits.log 2>&1; tail -4 gpurun_out/${TAG}_tests_$bits.log"
done
