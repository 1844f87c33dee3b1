#!/usr/bin/env python3
"""Print a compact per-kernel table from a rocprofv3 *_kernel_stats.csv."""
import csv, re, sys, glob
path = sys.argv[1]
if "*" in path: path = sorted(glob.glob(path))[-1]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 14
for r in list(csv.DictReader(open(path)))[:top]:
    m = re.search(r'(k_\w+(<[^>]*>)?|__amd\w+)', r['Name'])
    print("%-30s calls=%-4s avg_us=%-9.1f total_ms=%-8.2f %5.1f%%" % (m.group(1) if m else r['Name'][:30], r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e6, float(r['Percentage'])))
