timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -1
for r in 1 2 3; do python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['avg_us'], d['roofline']['frac'], d['roofline_gemm']['avg_us'])"; done
bash tools/step_breakdown.sh 2>/dev/null | grep -i "atp_bwd\|atp_src\|skinny\|elu\|atp_fwd" | head
