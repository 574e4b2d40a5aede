#!/bin/bash
# in-step A/B of the x3p GEMM path and the side stream (inside gpurun)
for ws in 1 0; do for x in 0 1; do
  S2T_WGRAD_STREAM=$ws S2T_WHITEN_STREAM=$ws S2T_X3P=$x timeout -k 10 300 python bench.py --steps 10 --warmup 8 --no-cpu-baseline --profile-steps 0 > gpurun_out/ab_${ws}_${x}.json 2> gpurun_out/ab_${ws}_${x}.err
  python - <<PY
import json
d = json.load(open("gpurun_out/ab_${ws}_${x}.json"))
print("side", $ws, "x3p", $x, "ms", round(d["ms_per_step"], 2), d["config"]["gemm_paths"], d["roofline"]["kernel"], round(d["roofline"]["frac"] or 0, 3))
PY
done; done
S2T_PLAN_DUMP=1 S2T_X3P=1 timeout -k 10 300 python bench.py --steps 4 --warmup 6 --no-cpu-baseline --profile-steps 0 2>/dev/null | grep "s2t plan" > gpurun_out/plan_dump.log
wc -l gpurun_out/plan_dump.log; grep -c "x3p" gpurun_out/plan_dump.log
