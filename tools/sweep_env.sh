#!/bin/bash
# same-box sweep of one environment variable over the default bench (inside gpurun):
#   bash tools/sweep_env.sh NAME "v1 v2 ..." REPS
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for i in $(seq ${3:-2}); do
  for v in $2; do
    ms=$(env $1=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --profile-steps 0 2>/dev/null \
         | python -c "import json,sys;print(round(json.loads(sys.stdin.read())['ms_per_step'],2))")
    echo "$1=$v  $ms ms/step"
  done
done
