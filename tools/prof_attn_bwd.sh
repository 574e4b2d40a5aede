#!/bin/bash
# per-kernel averages of tools/bench_attn_bwd.py under rocprofv3 for each env setting given as args
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for s in "$@"; do
  rm -rf gpurun_out/ab_prof
  export $s
  rocprofv3 --kernel-trace --stats -d gpurun_out/ab_prof -o ab --output-format csv -- python tools/bench_attn_bwd.py > gpurun_out/ab.log 2>&1
  unset ${s%%=*}
  echo "== $s"
  python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/ab_prof/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:8]:
    if "attn" in r["Name"]:
        print(f"{float(r['AverageNs'])/1e3:9.1f} us x{r['Calls']:>4}  {r['Name'][:90]}")
PY
done
find gpurun_out/ab_prof -name "*_kernel_trace.csv" -delete
