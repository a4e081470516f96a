#!/bin/bash
# usage: tools/pmc_run.sh <tag> "<counters>" -- <program args>      (runs on the GPU box)
TAG=$1; CTRS=$2; shift 3
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $ROOT
rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d $OUT -o c -- "$@" > $OUT/log.txt 2>&1
python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"].replace("void recon::(anonymous namespace)::", "")[:60]
    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    if "at::" in k: continue
    print(k)
    for c, v in sorted(d.items()):
        print("    %-28s n=%3d avg=%16.1f" % (c, len(v), sum(v) / len(v)))
PY
