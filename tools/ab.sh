#!/bin/bash
# A/B of one environment switch on the same box: bash tools/ab.sh VAR A B [reps] [bench args]
# prints ms_per_step of `bench.py --no-cpu-baseline --profile-steps 0` alternating VAR=A / VAR=B
VAR=$1; A=$2; B=$3; REPS=${4:-2}; shift 4
for i in $(seq $REPS); do
  for v in $A $B; do
    ms=$(env $VAR=$v python bench.py --steps 20 --warmup 4 --no-cpu-baseline --profile-steps 0 "$@" 2>/dev/null \
         | python -c "import json,sys;print(round(json.loads(sys.stdin.read())['ms_per_step'],2))")
    echo "$VAR=$v  $ms ms/step"
  done
done
