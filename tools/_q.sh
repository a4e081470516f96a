cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pl64 -o pl -- python3 tools/powerlaw_bench.py 64 > gpurun_out/pl64.log 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/pl64/**/*kernel_stats.csv",recursive=True)[0]
for i,r in enumerate(csv.DictReader(open(f))):
    if i<30: print(r["Name"][:100], r["Calls"], r["AverageNs"], r["Percentage"])
PY
