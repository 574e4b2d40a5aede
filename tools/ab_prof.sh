#!/bin/bash
# rocprofv3 per-kernel summary of the C3 bench for several source trees on ONE box:
#   bash tools/ab_prof.sh dirA dirB ...   -> gpurun_out/abp_<n>_timed_region.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
n=0
for d in "$@"; do
  n=$((n+1)); TAG=abp_$n
  rm -rf gpurun_out/${TAG}_prof
  (cd $d && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof -o ${TAG} --output-format csv -- python bench.py --steps 5 --warmup 3 --no-cpu-baseline --profile-steps 0 $BENCH_ARGS > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_profbench.json 2> $GRAFT_REPO_ROOT/gpurun_out/${TAG}_profbench.err)
  MS=$(python -c "import json;print(json.load(open('gpurun_out/${TAG}_profbench.json'))['ms_per_step'])")
  python tools/prof_summary.py gpurun_out/${TAG}_prof ${TAG} 5 $MS
  cp profiles/${TAG}_timed_region.txt gpurun_out/${TAG}_timed_region.txt
  rm -f profiles/${TAG}_*
  find gpurun_out/${TAG}_prof -name "*_kernel_trace.csv" -delete
  echo "== $d: $MS ms/step (profiled)"; head -8 gpurun_out/${TAG}_timed_region.txt | cut -c1-120
done
