#!/bin/bash
# rocprofv3 per-kernel averages of the attention micro-benchmark at T=495 (inside gpurun)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/attn_prof
rocprofv3 --kernel-trace --stats -d gpurun_out/attn_prof -o attn --output-format csv -- python tools/bench_kernels.py attn > gpurun_out/attn_bench.log 2>&1
cat gpurun_out/attn_bench.log | grep -v amdgpu
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/attn_prof/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in rows[:14]:
    print(f"{float(r['AverageNs'])/1e3:9.1f} us x{r['Calls']:>5}  {r['Name'][:110]}")
PY
find gpurun_out/attn_prof -name "*_kernel_trace.csv" -delete
