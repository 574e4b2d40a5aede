#!/bin/bash
# usage (inside gpurun): bash tools/sweep_env.sh VAR "v1 v2 ..." CONFIG ENTRY
# runs a short bench per value and prints ms/step + the entry's average launch time
VAR=$1; VALS=$2; CFG=${3:-C3}; ENTRY=${4:-s2t_gemm_tn_grouped}
mkdir -p gpurun_out
for v in $VALS; do
  env $VAR=$v python bench.py --config $CFG --steps 8 --warmup 3 --no-cpu-baseline --profile-steps 1 > gpurun_out/sweep_${VAR}_${v}_${CFG}.json 2> gpurun_out/sweep_${VAR}_${v}_${CFG}.err || { echo "$VAR=$v FAILED"; tail -5 gpurun_out/sweep_${VAR}_${v}_${CFG}.err; continue; }
  python - <<PY
import json
d=json.load(open("gpurun_out/sweep_${VAR}_${v}_${CFG}.json"))
k=[x for x in d["roofline"]["kernels"] if x["entry"]=="$ENTRY"]
print("$VAR=$v $CFG: %.2f ms/step  %s avg %.1f us x %.0f  loss %.4f" % (d["ms_per_step"], "$ENTRY", k[0]["avg_us"] if k else -1, k[0]["launches_per_step"] if k else 0, d["config"]["final_loss"]))
PY
done
