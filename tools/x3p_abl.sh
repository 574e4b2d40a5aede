#!/bin/bash
export S2T_DEBUG_KERNELS=1   # the ablation switches are ignored without it
# ablations of the x3p main loop on one shape (inside gpurun): bash tools/x3p_abl.sh M N K tile [list]
LIST=${5:-"0 1 2 3 4 7 8 16 17 24 23"}
for a in $LIST; do
  echo "ABL $a: $(S2T_X3P_ABL=$a python tools/x3p_stamps.py $1 $2 $3 $4 2>&1 | grep -v amdgpu | grep -E 'per launch|main\(tile|epilogue issue' | tr '\n' ' ' | sed -e 's/  */ /g' -e 's/workgroups, span [0-9]* cycles//' -e 's/one CU.*//' | cut -c1-230)"
done
