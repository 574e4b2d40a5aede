#!/bin/bash
# round 6, visit j: the frontend alone -- time and kernel trace
TAG=${1:-r06j}
mkdir -p gpurun_out
export TAG
. tools/gpu_step.sh
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
step fe bash -c "timeout -k 10 300 python tools/bench_frontend.py 2>&1 | tail -2 | tee gpurun_out/${TAG}_fe.txt"
rm -rf gpurun_out/${TAG}_feprof
step feprof bash -c "FE_ITERS=5 rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_feprof -o fe --output-format csv -- python tools/bench_frontend.py > gpurun_out/${TAG}_feprof.log 2>&1; tail -2 gpurun_out/${TAG}_feprof.log"
python - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/%s_feprof/**/*kernel_trace.csv" % __import__("os").environ["TAG"], recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last 5 iterations: kernels after the 8 warm-up iterations; take the last 5/13 of launches by count
n = len(rows)
per = n // 13
last = rows[n - 5 * per:]
agg = collections.defaultdict(lambda: [0, 0.0])
for r in last:
    k = r["Kernel_Name"]
    agg[k][0] += 1
    agg[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
tot = sum(v[1] for v in agg.values()) / 5
out = open("gpurun_out/%s_fe_kernels.txt" % __import__("os").environ["TAG"], "w")
print("frontend alone: %.2f ms of kernel time per iteration, %d launches" % (tot / 1e3, per), file=out)
for k, (c, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print("%8.1f us/iter %5.1f calls  %s" % (us / 5, c / 5, k[:150]), file=out)
out.close()
print(open("gpurun_out/%s_fe_kernels.txt" % __import__("os").environ["TAG"]).read())
PY
find gpurun_out/${TAG}_feprof -name "*_kernel_trace.csv" -delete
