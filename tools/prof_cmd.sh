#!/bin/bash
# rocprofv3 per-kernel averages of an arbitrary python tool: bash tools/prof_cmd.sh PATTERN tool.py [args]   (inside gpurun)
PAT=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/cmd_prof
rocprofv3 --kernel-trace --stats -d gpurun_out/cmd_prof -o cmd --output-format csv -- python "$@" > gpurun_out/cmd.log 2>&1
PAT="$PAT" python - <<'PY'
import csv, glob, os
f = glob.glob("gpurun_out/cmd_prof/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if os.environ["PAT"] in r["Name"]:
        print(f"{float(r['AverageNs'])/1e3:9.1f} us x{r['Calls']:>5}  {r['Name'][:100]}")
PY
find gpurun_out/cmd_prof -name "*_kernel_trace.csv" -delete
