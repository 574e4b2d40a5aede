#!/bin/bash
# Same-box A/B of several source trees: bash tools/ab_dirs.sh REPS dir1 dir2 ... (bench args via BENCH_ARGS)
REPS=${1:-2}; shift
for i in $(seq $REPS); do
  for d in "$@"; do
    ms=$(cd $d && python bench.py --steps 20 --warmup 4 --no-cpu-baseline --profile-steps 0 $BENCH_ARGS 2>/dev/null \
         | python -c "import json,sys;print(round(json.loads(sys.stdin.read())['ms_per_step'],2))")
    echo "$d  $ms ms/step"
  done
done
