#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace + stats of bench.py, then two separate
# PMC passes (FETCH_SIZE, WRITE_SIZE) as /opt/skills/guides/MI355X_MICROARCH.md prescribes.
# Outputs land in gpurun_out/prof_<tag>/ ; tools/prof_summary.py condenses them for profiles/.
set -u
TAG=${1:-r1}
ARGS=${2:---steps 10 --warmup 3 --no-cpu-baseline --no-extras}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 bench.py $ARGS > $OUT/bench_trace.log 2>&1
echo "trace rc=$?"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o f -- python3 bench.py $ARGS > $OUT/bench_fetch.log 2>&1
echo "fetch rc=$?"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o w -- python3 bench.py $ARGS > $OUT/bench_write.log 2>&1
echo "write rc=$?"
python3 tools/prof_summary.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt | head -60
# keep the merge-back small: the raw per-dispatch traces are large
find $OUT -name "*_kernel_trace.csv" -size +20M -delete
