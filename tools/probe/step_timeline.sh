#!/bin/bash
# ordered kernel timeline of ONE cfg 2 bench step (start offsets, durations, gaps between consecutive kernels); runs on the GPU box
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp && cd $ROOT
rm -rf gpurun_out/tl; rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -o t -- python3 bench.py --steps 12 --warmup 4 --no-extras --no-cpu-baseline > gpurun_out/tl.log 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/tl/**/*kernel_trace.csv",recursive=True)[0]
rows=sorted(csv.DictReader(open(f)), key=lambda r:int(r["Start_Timestamp"]))
def short(n):
    for p in ("void recon::(anonymous namespace)::","recon::(anonymous namespace)::","void (anonymous namespace)::","(anonymous namespace)::","void recon::"): n=n.replace(p,"")
    return n.split("(")[0][:60]
# find the last occurrences of the step's first kernel (k_score_vec) and print one full step from there
idx=[i for i,r in enumerate(rows) if short(r["Kernel_Name"]).startswith("k_score_vec") and not short(r["Kernel_Name"]).startswith("k_score_vec_bwd")]
# steps: k_score_vec is launched once per forward
starts=idx
if len(starts)>=12:
    a,b=starts[8],starts[10]
    t0=int(rows[a]["Start_Timestamp"]); prev=None; busy=0
    for r in rows[a:b]:
        s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
        gap=(s-prev)/1e3 if prev else 0.0
        busy+=(e-s)
        print("%-62s start %8.1f us  dur %7.1f  gap %6.1f"%(short(r["Kernel_Name"]),(s-t0)/1e3,(e-s)/1e3,gap)); prev=e
    print("step span %.1f us, busy %.1f us, launches %d"%((int(rows[b]["Start_Timestamp"])-t0)/1e3,busy/1e3,b-a))
PY
grep -o '"ms_per_step": [0-9.]*' gpurun_out/tl.log
rm -rf gpurun_out/tl
