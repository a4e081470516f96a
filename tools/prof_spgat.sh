#!/bin/bash
# per-kernel times of tools/spgat_bench.py (full SpGAT: 8 heads + out_att), runs on the GPU box
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp && cd $ROOT
rm -rf gpurun_out/prof_spgat; rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_spgat -o t -- python3 tools/spgat_bench.py > gpurun_out/prof_spgat.log 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/prof_spgat/**/*kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:26]:
    n=r["Name"].replace("void recon::(anonymous namespace)::","").replace("recon::(anonymous namespace)::","")[:70]
    print("%-72s %5s %10.1f %9.1f"%(n,r["Calls"],float(r["TotalDurationNs"])/1e3,float(r["AverageNs"])/1e3))
PY
find gpurun_out/prof_spgat -name "*_kernel_trace.csv" -delete
