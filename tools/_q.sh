cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/bd_prop -o t -- python3 tools/bench_prop.py > gpurun_out/bd_prop.log 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/bd_prop/**/*kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:22]:
    n=r["Name"].replace("void recon::(anonymous namespace)::","").replace("recon::(anonymous namespace)::","")[:80]
    print("%-82s %5s %9.1f %8.1f"%(n,r["Calls"],float(r["TotalDurationNs"])/1e3,float(r["AverageNs"])/1e3))
PY
