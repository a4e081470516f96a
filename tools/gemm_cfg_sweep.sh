#!/bin/bash
# Runs on the GPU box: per-kernel GEMM times of bench.py under each RECON_GEMM_CFG (rocprofv3 --stats).
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp && cd $ROOT
for cfg in ${1:-0 3 4}; do
  export RECON_GEMM_CFG=$cfg
  OUT=$ROOT/gpurun_out/sweep_cfg$cfg
  rm -rf $OUT; mkdir -p $OUT
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/log.txt 2>&1
  echo "== RECON_GEMM_CFG=$cfg  $(grep -o '"ms_per_step": [0-9.]*' $OUT/log.txt)"
  python3 - <<PY
import csv, glob
f = glob.glob("$OUT/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r["Name"].replace("void recon::(anonymous namespace)::", "")
    if "gemm" in n or "splitk" in n:
        print("   %-60s calls %4s avg %8.1f us" % (n[:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
  find $OUT -name "*_kernel_trace.csv" -delete
done
