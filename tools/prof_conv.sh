# rocprofv3 kernel times of the conv-module kernels at the C3 shapes, for several S2T_CONV_BLOCKS
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for blocks in ${@:-384}; do
rm -rf gpurun_out/pc_$blocks
S2T_CONV_BLOCKS=$blocks rocprofv3 --kernel-trace --stats -d gpurun_out/pc_$blocks -o r --output-format csv -- python tools/bench_kernels.py conv > /dev/null 2>&1
echo "== S2T_CONV_BLOCKS=$blocks"; f=$(find gpurun_out/pc_$blocks -name "*kernel_stats.csv" | head -1); python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n=r['Name']
    if 'zipconv' in n:
        print(f"{n[:70]:70s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.1f} us  min {float(r['MinNs'])/1e3:8.1f} max {float(r['MaxNs'])/1e3:8.1f}")
PY
find gpurun_out/pc_$blocks -name "*_kernel_trace.csv" -delete
done
