# two processes on ONE GPU over gloo: exercises bench.py's --gpus N path (collective ordering, cfg4 / cfg5 workloads); not a performance figure
export RECON_DIST_BACKEND=gloo
for W in cfg4 cfg5; do
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 5 --warmup 2 --workload $W --graphs 128 --no-cpu-baseline 2>&1 | grep '"metric"' | cut -c1-700
  RECON_DP_OVERLAP=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 5 --warmup 2 --workload $W --graphs 128 --no-cpu-baseline 2>&1 | grep '"metric"' | cut -c1-400
done
python bench.py --workload cfg5 --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>&1 | grep '"metric"' | cut -c1-900
