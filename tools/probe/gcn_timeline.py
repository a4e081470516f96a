"""Kernel timeline of the last gcn_stack() training iterations in a rocprofv3 kernel trace (t_kernel_trace.csv)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'stack_bwd' in r['Kernel_Name']]
for which in (-6, -3):
    i = idx[which]
    j = i
    while 'pad_both_multi' not in rows[j]['Kernel_Name']: j -= 1
    k = i
    while 'reduce_multi' not in rows[k]['Kernel_Name']: k += 1
    prev = None; tot = 0
    for r in rows[j - 1:k + 2]:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        name = r['Kernel_Name'].split('(')[0].replace('recon::', '').replace('anonymous namespace', '')[-44:]
        print(name.ljust(46), 'dur %6.1f' % ((e - s) / 1e3), 'gap %6.1f' % (((s - prev) / 1e3) if prev else 0))
        prev = e
    print()
