#!/bin/bash
# per-kernel average times of the bench step (runs on the GPU box)
TAG=${1:-x}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp && cd $ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/bd_$TAG -o t -- python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-extras > gpurun_out/bd_$TAG.log 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/bd_$TAG/**/*kernel_stats.csv",recursive=True)[0]
tot=0
for r in list(csv.DictReader(open(f)))[:22]:
    n=r["Name"].replace("void recon::(anonymous namespace)::","").replace("recon::(anonymous namespace)::","")[:64]
    print("%-66s %5s %9.1f %8.1f"%(n,r["Calls"],float(r["TotalDurationNs"])/1e3,float(r["AverageNs"])/1e3))
PY
grep -o '"ms_per_step": [0-9.]*' gpurun_out/bd_$TAG.log
