#!/bin/bash
# launch census of a fresh stage-A iteration per phase (runs on the GPU box): rocprofv3 kernel trace of tools/stage_a_census.py, split at its sentinel fills
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp && cd $ROOT
rm -rf gpurun_out/census; rocprofv3 --kernel-trace --output-format csv -d gpurun_out/census -o t -- python3 tools/stage_a_census.py --iters 5 > gpurun_out/census.log 2>&1
F=$(find gpurun_out/census -name "*kernel_trace.csv" | head -1)
python3 tools/stage_a_census.py --parse $F --top ${1:-14} > gpurun_out/stage_a_census.json
python3 -c "
import json; d=json.load(open('gpurun_out/stage_a_census.json'))
for k,v in d.items():
    if isinstance(v,dict):
        print('%-18s %6.1f'%(k, v['launches_per_iteration']))
        for n,c in v['top']: print('     %5.1f  %s'%(c,n))
    else: print(k,v)"
find gpurun_out/census -name "*kernel_trace.csv" -delete
