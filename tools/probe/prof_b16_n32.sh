#!/bin/bash
# rocprofv3 kernel stats of the bfloat16 n = 32 propagation (tools/b16_quick.py 32); runs on the GPU box
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp && cd $ROOT
rm -rf gpurun_out/prof_b16; rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_b16 -o t -- python3 tools/b16_quick.py 32 > gpurun_out/prof_b16.log 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/prof_b16/**/*kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:12]:
    n=r["Name"].replace("void recon::(anonymous namespace)::","").replace("recon::(anonymous namespace)::","")[:80]
    print("%-82s %6s %10.1f %9.1f"%(n,r["Calls"],float(r["TotalDurationNs"])/1e3,float(r["AverageNs"])/1e3))
PY
find gpurun_out/prof_b16 -name "*_kernel_trace.csv" -delete
