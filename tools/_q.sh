cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -x -q -m gpu -k "rowsum or spmm or spgat or spkbgat or indexed" 2>&1 | tail -3
python3 tools/kg_scale_bench.py --kbgat
python3 tools/kg_scale_bench.py | cut -c90-400
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kb -o kb -- python3 tools/kg_scale_bench.py --kbgat > gpurun_out/kb.log 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/kb/**/*kernel_stats.csv",recursive=True)[0]
for i,r in enumerate(csv.DictReader(open(f))):
    if i < 12 or "rowsum" in r["Name"]: print(r["Name"][:100], r["Calls"], r["AverageNs"], r["Percentage"])
PY
