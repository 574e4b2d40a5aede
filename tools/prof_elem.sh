# rocprofv3 kernel times for tools/bench_elem.py (the script itself is host-bound)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pe
rocprofv3 --kernel-trace --stats -d gpurun_out/pe -o r --output-format csv -- python tools/bench_elem.py > /dev/null 2>&1
f=$(find gpurun_out/pe -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
# group consecutive by kernel name keeping order of first appearance & grid size
agg=collections.OrderedDict()
for r in rows:
    n=r['Kernel_Name']
    if not any(k in n for k in ('biasnorm_bwd','bypass','downsample')): continue
    key=(n[:60], r.get('Grid_Size_X') or r.get('Grid_Size'), r.get('Workgroup_Size_X') or '')
    d=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
    agg.setdefault(key,[]).append(d)
for k,v in agg.items():
    v=sorted(v)
    print(f"{k[0]:60s} grid {k[1]:>8s} n={len(v):3d} median {v[len(v)//2]:7.1f} us")
PY
find gpurun_out/pe -name "*_kernel_trace.csv" -delete
