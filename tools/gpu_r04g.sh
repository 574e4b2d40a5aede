#!/bin/bash
# End-of-round-4 measurement visit (inside gpurun): bash tools/gpu_r04g.sh
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
TAG=${1:-r04g}
L=gpurun_out/$TAG.log; : > $L
run() {  # label args...
  local lab="$1"; shift
  python bench.py "$@" --no-cpu-baseline > gpurun_out/${TAG}_tmp.json 2> gpurun_out/${TAG}_tmp.err
  python -c "
import json
ls=[l for l in open('gpurun_out/${TAG}_tmp.json') if l.startswith('{')]
if ls:
    d=json.loads(ls[-1]);print('$lab -> %.2f ms/step %.0f audio-s/s' % (d['ms_per_step'], d['value']))
else:
    print('$lab -> FAILED:', open('gpurun_out/${TAG}_tmp.err').read()[-300:].replace(chr(10),' | '))" >> $L
  tail -1 $L
}
run "default C3" --steps 20 --warmup 5
run "--ddp-force allreduce" --steps 10 --warmup 5 --ddp-force allreduce
run "--ddp-force rs_ag" --steps 10 --warmup 5 --ddp-force rs_ag
run "--accum 20" --steps 20 --warmup 5 --accum 20
run "--random-chunk" --steps 12 --warmup 6 --random-chunk
run "--config C2" --steps 10 --warmup 5 --config C2
run "--config C4" --steps 10 --warmup 5 --config C4
run "--config C5" --steps 10 --warmup 5 --config C5
python tools/exp_host.py 2>&1 | grep -E "single step|back to back" >> $L
bash tools/gpu_r04c.sh $TAG >> $L 2>&1
bash tools/gpu_pmc2.sh $TAG C3 >> gpurun_out/${TAG}_pmc.log 2>&1
tail -30 gpurun_out/${TAG}_pmc.log >> $L
