#!/bin/bash
# same-box A/B of one environment switch on the default bench: bash tools/ab_env.sh VAR A B [ROUNDS] [bench args]
VAR=$1; A=$2; B=$3; R=${4:-3}; shift 4
for i in $(seq 1 $R); do
  for V in $A $B; do
    MS=$(env $VAR=$V python bench.py --steps 20 --warmup 5 --no-cpu-baseline --profile-steps 0 "$@" 2>/dev/null | python -c "import sys,json; print('%.2f' % json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
    echo "$VAR=$V  $MS ms/step"
  done
done
