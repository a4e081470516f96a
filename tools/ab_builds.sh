#!/bin/bash
# A/B of two BUILDS of the library on one box: tools/ab_builds.sh "<EXTRA flags of build B>" [rounds] -- the in-tree .so is build A,
# build B is compiled HERE (build container) into recon_amd/csrc/librecon_hip_b.so and travels with the snapshot; on the GPU box:
#   bash tools/ab_builds.sh --run [rounds]      interleaved `bench.py --steps 200` of A and B, ms_per_step of each
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
if [ "$1" != "--run" ]; then
  EXTRA="$1"
  rm -rf /tmp/ab_b && mkdir -p /tmp/ab_b && cp $ROOT/recon_amd/csrc/*.hip $ROOT/recon_amd/csrc/*.h $ROOT/recon_amd/csrc/Makefile /tmp/ab_b/
  mkdir -p /tmp/include_ab && cp -r $ROOT/include /tmp/
  (cd /tmp/ab_b && sed -i "s|../../include|$ROOT/include|g" Makefile && make -j6 EXTRA="$EXTRA" > /tmp/ab_b/build.log 2>&1) || { tail -20 /tmp/ab_b/build.log; exit 1; }
  cp /tmp/ab_b/librecon_hip.so $ROOT/recon_amd/csrc/librecon_hip_b.so
  echo "build B ($EXTRA): recon_amd/csrc/librecon_hip_b.so"
  exit 0
fi
R=${2:-3}
for i in $(seq $R); do
  A=$(python3 $ROOT/bench.py --steps 200 --warmup 20 --no-extras --no-cpu-baseline | python3 -c "import sys,json; print('%.4f' % json.loads(sys.stdin.readline())['ms_per_step'])")
  B=$(RECON_HIP_LIB=$ROOT/recon_amd/csrc/librecon_hip_b.so python3 $ROOT/bench.py --steps 200 --warmup 20 --no-extras --no-cpu-baseline | python3 -c "import sys,json; print('%.4f' % json.loads(sys.stdin.readline())['ms_per_step'])")
  echo "round $i: A $A  B $B"
done
