#!/bin/bash
# rocprofv3 kernel trace of one python tool, summarised over its LAST iterations (a marker kernel
# that runs once per iteration delimits them; tuning launches of the first iterations stay out).
# usage (inside gpurun): bash tools/prof_one.sh TAG MARKER LAST tools/bench_frontend.py [args]
TAG=$1; MARK=$2; LAST=$3; shift 3
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
rm -rf gpurun_out/${TAG}_prof
rocprofv3 --kernel-trace -d gpurun_out/${TAG}_prof -o ${TAG} --output-format csv -- python3 "$@" > gpurun_out/${TAG}_prof.out 2>&1
python3 - <<PY
import csv,glob,collections
f=glob.glob("gpurun_out/${TAG}_prof/**/*kernel_trace.csv",recursive=True)[0]
tr=list(csv.DictReader(open(f)))
tr.sort(key=lambda r:int(r["Start_Timestamp"]))
marks=[i for i,r in enumerate(tr) if "${MARK}" in r["Kernel_Name"]]
n=int(${LAST})
lo=marks[-n-1]; hi=marks[-1]
sel=tr[lo:hi]
agg=collections.defaultdict(lambda:[0,0])
for r in sel:
    a=agg[r["Kernel_Name"]]; a[0]+=int(r["End_Timestamp"])-int(r["Start_Timestamp"]); a[1]+=1
tot=sum(a[0] for a in agg.values())
span=(int(sel[-1]["End_Timestamp"])-int(sel[0]["Start_Timestamp"]))/n/1e6
out=["%d iterations: kernel time %.3f ms/it, wall %.3f ms/it, %.1f launches/it"%(n,tot/n/1e6,span,len(sel)/n)]
for k,a in sorted(agg.items(), key=lambda kv:-kv[1][0])[:45]:
    out.append("%8.3f ms/it %6.1f calls/it %9.1f us  %s"%(a[0]/n/1e6, a[1]/n, a[0]/a[1]/1e3, k[:120]))
open("gpurun_out/${TAG}_kernels.txt","w").write("\n".join(out)+"\n")
print("\n".join(out))
PY
find gpurun_out/${TAG}_prof -name "*_kernel_trace.csv" -delete
