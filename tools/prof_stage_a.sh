#!/bin/bash
# per-kernel times of tools/stage_a_iter_bench.py (runs on the GPU box)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp && cd $ROOT
rm -rf gpurun_out/prof_stage_a; rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_stage_a -o t -- python3 tools/stage_a_iter_bench.py --iters 20 --loss-rows recon > gpurun_out/prof_stage_a.log 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/prof_stage_a/**/*kernel_stats.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time %.1f ms, %d kernels, %d launches"%(tot/1e6,len(rows),sum(int(r["Calls"]) for r in rows)))
for r in rows[:40]:
    n=r["Name"].replace("void recon::(anonymous namespace)::","").replace("recon::(anonymous namespace)::","")[:90]
    print("%-92s %6s %10.1f %9.1f"%(n,r["Calls"],float(r["TotalDurationNs"])/1e3,float(r["AverageNs"])/1e3))
PY
find gpurun_out/prof_stage_a -name "*_kernel_trace.csv" -delete
