#!/bin/bash
# helper for the per-visit scripts: run one step, record its exit code, and STOP the whole visit when
# a step was killed or timed out (no further GPU step after a hang)
step() {
  local name=$1; shift
  echo "=== $name: $*" >> gpurun_out/${TAG}_steps.log
  "$@"
  local rc=$?
  echo "=== $name rc=$rc" | tee -a gpurun_out/${TAG}_steps.log
  if [ $rc -eq 124 ] || [ $rc -ge 128 ]; then
    echo "step $name was killed / timed out: stopping the visit" | tee -a gpurun_out/${TAG}_steps.log
    exit $rc
  fi
  return 0
}
