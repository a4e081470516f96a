python -m pytest tests -x -q -m gpu -k "graph or hub or fuzz" 2>&1 | tail -3
for i in 1 2; do
python3 bench.py --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('defer', d['ms_per_step'], d['uncached']['ms_per_step'], d['uncached']['trusted_edges_ms_per_step'])"
python3 -c "
import sys; sys.argv=['bench.py','--no-cpu-baseline']
import recon_amd.graph as G; G.DEFER_HUB_READ=False
import runpy; runpy.run_path('bench.py', run_name='__main__')" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('eager', d['ms_per_step'], d['uncached']['ms_per_step'], d['uncached']['trusted_edges_ms_per_step'])"
done
