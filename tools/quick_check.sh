mkdir -p gpurun_out/$1
timeout 300 python -m pytest tests/test_gat_gpu.py -m gpu -x -q -k "heads_vs_oracle or golden or training_edge or power_law" > gpurun_out/$1/pytest.log 2>&1; tail -2 gpurun_out/$1/pytest.log
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/$1/bench.log 2>&1; tail -1 gpurun_out/$1/bench.log | cut -c1-330
bash tools/step_breakdown.sh $1 > gpurun_out/$1/bd.log 2>&1; head -12 gpurun_out/$1/bd.log
