#!/bin/bash
# Same-box A/B of the working tree against the copy of an earlier commit under .ab_prev/
# (made with: git archive <rev> | tar -x -C .ab_prev && (cd .ab_prev && python -m speech2text_amd.csrc.build))
REPS=${1:-2}
for i in $(seq $REPS); do
  for d in . .ab_prev; do
    ms=$(cd $d && python bench.py --steps 20 --warmup 4 --no-cpu-baseline --profile-steps 0 2>/dev/null \
         | python -c "import json,sys;print(round(json.loads(sys.stdin.read())['ms_per_step'],2))")
    echo "$d  $ms ms/step"
  done
done
