#!/bin/bash
# same-box A/B of environment settings: bash tools/ab_env.sh REPS "A=1" "A=0 B=2" ...  (bench args via BENCH_ARGS)
REPS=${1:-2}; shift
for i in $(seq $REPS); do
  for s in "$@"; do
    ms=$(env $s python bench.py --steps 20 --warmup 4 --no-cpu-baseline --profile-steps 0 $BENCH_ARGS 2>/dev/null \
         | python -c "import json,sys;print(round(json.loads(sys.stdin.read())['ms_per_step'],2))")
    echo "$s  $ms ms/step"
  done
done
