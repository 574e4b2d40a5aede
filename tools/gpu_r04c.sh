#!/bin/bash
# rocprof of the bench + per-entry table (inside gpurun): bash tools/gpu_r04c.sh TAG
set -o pipefail
TAG=${1:-r04c}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python bench.py --steps 10 --warmup 6 --no-cpu-baseline > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
python - <<PY
import json
d = json.load(open("gpurun_out/${TAG}_bench.json"))
print("ms/step", round(d["ms_per_step"], 2), d["config"]["gemm_paths"], d["roofline"]["kernel"], d["roofline"]["frac"])
for k in d["roofline"]["kernels"][:45]:
    print("%-30s %7.1f /step %7.3f ms  %s %s" % (k["entry"], k["launches_per_step"], k["ms_per_step"],
          ("hbm %.2f" % k["frac_hbm"]) if "frac_hbm" in k else "", ("mfma %.2f" % k["frac_mfma"]) if "frac_mfma" in k else ""))
PY
rm -rf gpurun_out/${TAG}_prof
rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_prof -o ${TAG} --output-format csv -- python bench.py --steps 5 --warmup 6 --no-cpu-baseline --profile-steps 0 > gpurun_out/${TAG}_profbench.json 2> gpurun_out/${TAG}_profbench.err
MS=$(python -c "import json;print(json.load(open('gpurun_out/${TAG}_profbench.json'))['ms_per_step'])")
python tools/prof_summary.py gpurun_out/${TAG}_prof ${TAG} 5 $MS
mkdir -p gpurun_out/profiles_${TAG} && cp profiles/${TAG}_* gpurun_out/profiles_${TAG}/
find gpurun_out/${TAG}_prof -name "*_kernel_trace.csv" -delete
head -10 profiles/${TAG}_timed_region.txt
