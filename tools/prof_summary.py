#!/usr/bin/env python3
"""Condense the rocprofv3 outputs of tools/profile_bench.sh: per-kernel time table (from the
kernel-stats CSV) and per-launch HBM traffic of the recon kernels from the FETCH_SIZE / WRITE_SIZE
PMC passes.  Units/corrections per /opt/skills/guides/MI355X_MICROARCH.md (HBM section): the
counters are in KiB; on gfx950 FETCH_SIZE reports half of the bytes of wide coalesced streaming
reads, so the read side is doubled; WRITE_SIZE is taken as reported (uncalibrated)."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def short(n):
    n = n.replace("void recon::(anonymous namespace)::", "").replace("recon::(anonymous namespace)::", "")
    n = n.replace("(anonymous namespace)::", "")
    return n.split("(")[0][:64]


stats = glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True)
if stats:
    rows = list(csv.DictReader(open(stats[0])))
    print("== kernel time (rocprofv3 --kernel-trace --stats), %s" % os.path.relpath(stats[0], out))
    print("%-66s %6s %12s %10s %10s %10s %7s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct"))
    for r in rows[:24]:
        print("%-66s %6s %12.1f %10.1f %10.1f %10.1f %7.2f" % (short(r["Name"]), r["Calls"], float(r["TotalDurationNs"]) / 1e3,
              float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, float(r["Percentage"])))

for tag, ctr in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    files = glob.glob(os.path.join(out, tag, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        continue
    acc = defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(files[0])):
        if r.get("Counter_Name") != ctr:
            continue
        k = short(r["Kernel_Name"])
        acc[k][0] += 1
        acc[k][1] += float(r["Counter_Value"])
    print("\n== %s per launch (KiB as reported -> MB; FETCH doubled per the gfx950 correction)" % ctr)
    for k, (n, v) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:12]:
        per = v / n * 1024.0
        if ctr == "FETCH_SIZE":
            print("%-66s launches=%5d  reported=%9.1f MB  corrected(x2)=%9.1f MB" % (k, n, per / 1e6, 2 * per / 1e6))
        else:
            print("%-66s launches=%5d  reported=%9.1f MB" % (k, n, per / 1e6))
