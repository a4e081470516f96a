#!/usr/bin/env python3
"""Copy the condensed rocprofv3 outputs of tools/profile_bench.sh / tools/profile_secondary.sh from gpurun_out/ (scratch) into profiles/
(tracked) under per-round names, and derive profiles/roundN_pmc_traffic.json (per-launch HBM bytes of the cfg-2 kernels: FETCH_SIZE x 2 —
the gfx950 correction of /opt/skills/guides/MI355X_MICROARCH.md — + WRITE_SIZE) that bench.py quotes in `roofline.traffic`.

  python tools/collect_profiles.py 3 gpurun_out/prof_r3a gpurun_out/prof_r3sec
"""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd, bench_dir = sys.argv[1], sys.argv[2]
sec_dir = sys.argv[3] if len(sys.argv) > 3 else None
P = os.path.join(ROOT, "profiles")


def short(n):
    for p in ("void recon::(anonymous namespace)::", "recon::(anonymous namespace)::", "void (anonymous namespace)::", "(anonymous namespace)::"):
        n = n.replace(p, "")
    return n.split("(")[0].split("<")[0]


shutil.copy(os.path.join(bench_dir, "summary.txt"), os.path.join(P, "round%s_bench_cfg2_summary.txt" % rnd))
ks = glob.glob(os.path.join(bench_dir, "trace", "**", "*kernel_stats.csv"), recursive=True)
if ks:
    shutil.copy(ks[0], os.path.join(P, "round%s_bench_cfg2_kernel_stats.csv" % rnd))
acc = defaultdict(lambda: {"fetch": [0, 0.0], "write": [0, 0.0]})
for tag, ctr, key in (("pmc_fetch", "FETCH_SIZE", "fetch"), ("pmc_write", "WRITE_SIZE", "write")):
    raw_rows = []
    for f in glob.glob(os.path.join(bench_dir, tag, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == ctr and "recon" in r["Kernel_Name"] or r.get("Counter_Name") == ctr and r["Kernel_Name"].startswith("k_"):
                a = acc[short(r["Kernel_Name"])][key]
                a[0] += 1
                a[1] += float(r["Counter_Value"])
                raw_rows.append((r.get("Dispatch_Id", ""), short(r["Kernel_Name"]), r.get("Grid_Size", ""), r.get("Workgroup_Size", ""), ctr, r["Counter_Value"]))
    # the raw counter dump behind the rounded JSON (one row per dispatch of a recon kernel, values as rocprofv3 reports them: KiB)
    with open(os.path.join(P, "round%s_pmc_%s_raw.csv" % (rnd, key)), "w") as fh:
        w = csv.writer(fh)
        w.writerow(("dispatch_id", "kernel", "grid_size", "workgroup_size", "counter", "value_KiB_as_reported"))
        w.writerows(raw_rows)
out = {"source": "tools/profile_bench.sh (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, bench.py --steps 10 --warmup 3 --no-extras), "
                 "profiles/round%s_bench_cfg2_summary.txt" % rnd,
       "correction": "FETCH_SIZE doubled (gfx950 reports half of wide coalesced reads, MI355X_MICROARCH.md HBM section); WRITE_SIZE as reported; KiB -> bytes"}
for k, v in sorted(acc.items()):
    fb = 2 * v["fetch"][1] / max(v["fetch"][0], 1) * 1024.0
    wb = v["write"][1] / max(v["write"][0], 1) * 1024.0
    out[k] = {"fetch_bytes": int(round(fb)), "write_bytes": int(round(wb)), "total_bytes": int(round(fb) + round(wb)),
              "launches": v["fetch"][0]}
json.dump(out, open(os.path.join(P, "round%s_pmc_traffic.json" % rnd), "w"), indent=1)
print("k_gat_atp_fwd:", out.get("k_gat_atp_fwd"))
if sec_dir:
    for w, name in (("secondary", "round%s_secondary_prop_gcn_powerlaw_summary.txt"), ("stage_a", "round%s_stage_a_iteration_summary.txt")):
        src = os.path.join(sec_dir, w + "_summary.txt")
        if os.path.exists(src):
            txt = open(src).read()
            lines = [ln for ln in txt.split("\n") if not ((ln.startswith("== W2") or ln.startswith("== E2") or ln.startswith("== workload: W2")) and "{" not in ln)]
            open(os.path.join(P, name % rnd), "w").write("\n".join(lines))
print(sorted(os.listdir(P)))
