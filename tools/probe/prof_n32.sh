#!/bin/bash
# rocprofv3 kernel stats of the float32 n = 32 training step (tools/probe/run_n32_fp32.py); runs on the GPU box
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp && cd $ROOT
rm -rf gpurun_out/prof_n32; rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_n32 -o t -- python3 tools/probe/run_n32_fp32.py ${1:-1024} > gpurun_out/prof_n32.log 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/prof_n32/**/*kernel_stats.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
for r in rows[:22]:
    n=r["Name"].replace("void recon::(anonymous namespace)::","").replace("recon::(anonymous namespace)::","")[:80]
    print("%-82s %6s %10.1f %9.1f"%(n,r["Calls"],float(r["TotalDurationNs"])/1e3,float(r["AverageNs"])/1e3))
PY
find gpurun_out/prof_n32 -name "*_kernel_trace.csv" -delete
