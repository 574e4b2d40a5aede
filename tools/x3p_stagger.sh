#!/bin/bash
# GPU: start-stagger sweep of the persistent x3p kernel (delay per CU slot, units of 64 cycles):
# bash tools/x3p_stagger.sh
cd $GRAFT_REPO_ROOT
for s in 0 150 300 600; do
  S2T_X3P_STAGGER=$s python tools/x3p_ab.py 2>&1 | grep -v amdgpu | grep -E "15872   768|15872   256   768|31680   512|sum of"
done
