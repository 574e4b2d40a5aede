#!/bin/bash
# same-box A/B of two builds of the library (ablib/lib_old.so, ablib/lib_new.so) on the default bench
R=${1:-3}; shift
for i in $(seq 1 $R); do
  for V in old new; do
    cp ablib/lib_$V.so speech2text_amd/libs2t_mi355.so
    MS=$(python bench.py --steps 20 --warmup 5 --no-cpu-baseline --profile-steps 0 "$@" 2>/dev/null | python -c "import sys,json; print('%.2f' % json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
    echo "lib_$V  $MS ms/step"
  done
done
