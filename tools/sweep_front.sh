#!/bin/bash
# same-box sweep of S2T_FRONT_W_SIDE (inside gpurun): bash tools/sweep_front.sh "7 6 3 0" REPS
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for i in $(seq ${2:-2}); do
  for v in $1; do
    ms=$(S2T_FRONT_W_SIDE=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --profile-steps 0 2>/dev/null \
         | python -c "import json,sys;print(round(json.loads(sys.stdin.read())['ms_per_step'],2))")
    echo "S2T_FRONT_W_SIDE=$v  $ms ms/step"
  done
done
