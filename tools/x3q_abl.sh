#!/bin/bash
export S2T_DEBUG_KERNELS=1   # the ablation switches are ignored without it
# GPU: ablation builds of the producer / consumer GEMM (wrong results; timing only):
#   bash tools/x3q_abl.sh
cd $GRAFT_REPO_ROOT
for a in ${X3Q_ABLS:-0 1 2 4 8 16 17 23 32 40 55 119}; do
  echo "QABL=$a"
  S2T_X3Q_ABL=$a X3P_CF=33 python tools/x3p_ab.py 2>&1 | grep -E "15872   768|31680   512|sum of" 
done
