#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp && cd $ROOT
rm -rf gpurun_out/k1_align; rocprofv3 --kernel-trace --output-format csv -d gpurun_out/k1_align -o t -- python3 tools/probe/k1_row_alignment.py
python3 - <<PY
import csv, glob
from collections import defaultdict
f = glob.glob("gpurun_out/k1_align/**/*kernel_trace.csv", recursive=True)[0]
runs = []
cur = None
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if "k_gat_atp_fwd" in n:
        runs.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
runs.sort()
# widths in order, 23 / 31 / 39 / 55 / 87 launches each (3 warm-up + 20 + F - 192)
at = 0
for F_ in (192, 200, 208, 224, 256):
    k = 3 + 20 + F_ - 192
    d = [x[1] for x in runs[at:at + k]][3:]
    at += k
    mb = 8192 * (8 * 2 * F_ + F_) * 4 / 1e6
    XX
PY
find gpurun_out/k1_align -name "*_kernel_trace.csv" -delete
