#!/bin/bash
# round 6, visit bw: the session-start tree (commit 68355bf, unpacked under _old/) against this one, same box
TAG=${1:-r06bw}
mkdir -p gpurun_out
export TAG
. tools/gpu_step.sh
ROOT=$(pwd)
b() {
  local name=$1; local dir=$2; local cfg=$3
  step bench_$name bash -c "cd $dir && timeout -k 10 600 python bench.py --config $cfg --steps 30 --warmup 5 --no-cpu-baseline --profile-steps 0 2>> $ROOT/gpurun_out/${TAG}_bench.err | tail -1 | python -c \"import json,sys; d=json.loads(sys.stdin.read()); print('$name', round(d['ms_per_step'],2))\" | tee -a $ROOT/gpurun_out/${TAG}_ab.txt"
}
for r in 1 2; do
for cfg in C3 C4 C5 C2; do
b new_${cfg}_$r $ROOT $cfg
b old_${cfg}_$r $ROOT/_old $cfg
done
done
