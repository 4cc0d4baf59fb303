#!/usr/bin/env python3
"""Print the kernel sequence (name:µs) of the last ilqr_solve in a rocprofv3 kernel trace csv."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows)
idx = [i for i, (s, e, n) in enumerate(ev) if 'cost_kernel' in n][-1]
ev = ev[idx:]
keys = ('backward', 'evaluate_listed', 'evaluate_kernel', 'commit', 'repack_move', 'repack_count', 'repack_scan', 'cost', 'fillBuffer')
tot = {}
line = []
for s, e, n in ev:
    k = next((k for k in keys if k in n), n[:20])
    d = (e - s) / 1e3
    tot[k] = tot.get(k, 0) + d
    if d > 8: line.append(f"{k[:9]}:{d:.0f}")
print(' '.join(line))
print({k: round(v / 1e3, 2) for k, v in tot.items()}, 'span ms', (ev[-1][1] - ev[0][0]) / 1e6)
