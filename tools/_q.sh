timeout 900 python -m pytest tests/test_gat_gpu.py -m gpu -x -q 2>&1 | tail -1
timeout 300 python tools/gemm_hx2_bench.py 2>&1 | grep "^tn" | cut -c1-30,95-125
for r in 1 2 3; do python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['avg_us'], d['roofline']['frac'], d['roofline_gemm']['avg_us'])"; done
