#!/bin/bash
# stage-A iteration under several builds of the library (RECON_HIP_LIB), interleaved:  bash tools/probe/lib_ab_stage_a.sh cur s64 s32
for r in 1 2 3; do for v in "$@"; do
  if [ $v = cur ]; then L=""; else L=$PWD/recon_amd/csrc/librecon_hip_$v.so; fi
  RECON_HIP_LIB=$L python tools/stage_a_iter_bench.py --loss-rows recon 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$v', round(d['iteration_cached_batch_ms'],3), round(d['iteration_fresh_batch_ms'],3))"
done; done
