#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel stats + FETCH_SIZE / WRITE_SIZE (separate passes, as MI355X_MICROARCH.md prescribes) of the
# secondary workloads (tools/secondary.py: cfg 3b propagation n = 9 / 32, cfg 3a GraphConvolution bf16, cfg 5 power-law SpGAT) and of the
# stage-A iteration (tools/stage_a_iter_bench.py).  Condensed into gpurun_out/prof_<tag>/*.txt for profiles/.
set -u
TAG=${1:-r3sec}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $ROOT
for W in secondary stage_a; do
  if [ $W = secondary ]; then CMD="tools/secondary.py"; else CMD="tools/stage_a_iter_bench.py --iters 20 --loss-rows recon"; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${W}_trace -o t -- python3 $CMD > $OUT/${W}_trace.log 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/${W}_fetch -o f -- python3 $CMD > $OUT/${W}_fetch.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/${W}_write -o w -- python3 $CMD > $OUT/${W}_write.log 2>&1
  python3 - <<PY > $OUT/${W}_summary.txt 2>&1
import csv, glob, os
from collections import defaultdict
out = "$OUT"; W = "$W"
def short(n):
    for p in ("void recon::(anonymous namespace)::", "recon::(anonymous namespace)::", "void (anonymous namespace)::", "(anonymous namespace)::", "void recon::"):
        n = n.replace(p, "")
    return n.split("(")[0][:70]
print("== workload: %s" % open(os.path.join(out, W + "_trace.log")).read().strip().replace("\n", "\n== ")[-3000:])
f = glob.glob(os.path.join(out, W + "_trace", "**", "*kernel_stats.csv"), recursive=True)
if f:
    rows = list(csv.DictReader(open(f[0])))
    print("\n== kernel time (rocprofv3 --kernel-trace --stats)")
    print("%-72s %6s %12s %10s %10s %10s %7s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct"))
    for r in rows[:32]:
        print("%-72s %6s %12.1f %10.1f %10.1f %10.1f %7.2f" % (short(r["Name"]), r["Calls"], float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3,
              float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, float(r["Percentage"])))
for tag, ctr in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    files = glob.glob(os.path.join(out, W + "_" + tag, "**", "*counter_collection.csv"), recursive=True)
    if not files: continue
    acc = defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(files[0])):
        if r.get("Counter_Name") != ctr: continue
        k = short(r["Kernel_Name"]) + " grid=" + r.get("Grid_Size", "?")
        acc[k][0] += 1; acc[k][1] += float(r["Counter_Value"])
    print("\n== %s per launch (KiB as reported -> MB%s)" % (ctr, "; corrected = x2, the gfx950 rule for wide coalesced reads" if ctr == "FETCH_SIZE" else ""))
    for k, (n, v) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:18]:
        per = v / n * 1024.0
        print("%-84s launches=%5d  reported=%9.1f MB%s" % (k, n, per / 1e6, ("  corrected(x2)=%9.1f MB" % (2 * per / 1e6)) if ctr == "FETCH_SIZE" else ""))
PY
  find $OUT -name "*_kernel_trace.csv" -delete
  find $OUT -name "*counter_collection.csv" -size +30M -delete
done
head -60 $OUT/secondary_summary.txt
