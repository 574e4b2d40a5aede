#!/bin/bash
# the C3 bench variants on one box (inside gpurun): bash tools/gpu_variants.sh TAG
TAG=${1:-var}
run() { python bench.py --steps 20 --warmup 5 --no-cpu-baseline --profile-steps 0 "$@" 2>/dev/null | python -c "import sys,json; print('%.2f' % json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"; }
{
echo "default            $(run) ms/step"
echo "--random-chunk     $(run --random-chunk) ms/step"
echo "--accum 20         $(run --accum 20) ms/step"
echo "--ddp-force allreduce $(run --ddp-force allreduce) ms/step"
echo "--ddp-force rs_ag  $(run --ddp-force rs_ag) ms/step"
echo "--batch 16         $(run --batch 16) ms/step"
echo "--batch 32         $(run --batch 32) ms/step"
echo "default again      $(run) ms/step"
} > gpurun_out/${TAG}_variants.txt 2>&1
cat gpurun_out/${TAG}_variants.txt
