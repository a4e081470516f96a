cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/bd_spgat -o t -- python3 tools/spgat_bench.py 200 > gpurun_out/bd_spgat.log 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/bd_spgat/**/*kernel_stats.csv",recursive=True)[0]
tot=0
for r in list(csv.DictReader(open(f)))[:40]:
    n=r["Name"].replace("void recon::(anonymous namespace)::","").replace("recon::(anonymous namespace)::","")[:70]
    print("%-72s %5s %9.1f %8.1f"%(n,r["Calls"],float(r["TotalDurationNs"])/13e3,float(r["AverageNs"])/1e3))
PY
