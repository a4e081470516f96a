#!/bin/bash
# per-kernel times of the two cfg 5 legs of tools/secondary.py (runs on the GPU box): tools/prof_cfg5.sh <tag>
TAG=${1:-x}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp && cd $ROOT
rm -rf gpurun_out/cfg5_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/cfg5_$TAG -o t -- python3 tools/probe/run_cfg5_both.py > gpurun_out/cfg5_$TAG.log 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/cfg5_$TAG/**/*kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:30]:
    n=r["Name"].replace("void recon::(anonymous namespace)::","").replace("recon::(anonymous namespace)::","")[:64]
    print("%-66s %5s %9.1f %8.1f"%(n,r["Calls"],float(r["TotalDurationNs"])/1e3,float(r["AverageNs"])/1e3))
PY
grep -o '"fwd_bwd_ms": [0-9.]*' gpurun_out/cfg5_$TAG.log
find gpurun_out/cfg5_$TAG -name "*_kernel_trace.csv" -delete
