timeout 900 python -m pytest tests/test_gat_gpu.py -m gpu -x -q 2>&1 | tail -1
for w in 4 8; do echo "KM_WAVES=$w"; RECON_HX2_KM_WAVES=$w timeout 300 python tools/gemm_hx2_bench.py 2>&1 | grep "^tn" | cut -c1-30,95-125; done
for r in 1 2; do for w in 4 8; do echo -n "KM_WAVES=$w: "; RECON_HX2_KM_WAVES=$w python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['avg_us'], d['roofline']['frac'], d['roofline_gemm']['avg_us'])"; done; done
