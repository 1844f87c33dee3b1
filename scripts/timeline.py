#!/usr/bin/env python3
"""Print the kernel timeline of one steady-state frame from a rocprofv3 *_kernel_trace.csv."""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
seq = []
for r in rows:
    m = re.search(r'(k_\w+(<[^>]*>)?|__amd\w+)', r['Kernel_Name'])
    seq.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), m.group(1) if m else r['Kernel_Name'][:20], r.get('Queue_Id')))
seq.sort()
idx = [i for i, s in enumerate(seq) if s[2] in ('k_init_spawn', 'k_init_nodes')]
i0, i1 = idx[len(idx) // 2], idx[len(idx) // 2 + 1]   # a steady-state frame from the middle of the run
t0 = seq[i0][0]
prev = None
for s in seq[i0 - 1:i1]:
    gap = (s[0] - prev) / 1e3 if prev else 0.0
    print("%-28s q=%s start=%8.1f dur=%7.1f gap=%5.1f" % (s[2], s[3], (s[0] - t0) / 1e3, (s[1] - s[0]) / 1e3, gap))
    prev = s[1]
