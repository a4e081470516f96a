#!/usr/bin/env python3
"""Per (kernel, grid size) averages of a rocprofv3 --pmc counter_collection.csv (one kernel name can be several shapes)."""
import csv, sys, collections, glob
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
pat = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    if pat in r["Kernel_Name"]:
        k = (r["Kernel_Name"].replace("void recon::(anonymous namespace)::", "").replace("recon::(anonymous namespace)::", "")[:28], r["Grid_Size"])
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print("    %-30s n=%3d avg=%16.1f" % (c, len(v), sum(v) / len(v)))
