#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: per kernel name and counter, the mean
per dispatch.  Usage: pmc_summary.py out.txt file1.csv file2.csv ..."""
import csv, sys, collections, re
out = open(sys.argv[1], "w")
for path in sys.argv[2:]:
    acc = collections.defaultdict(lambda: [0.0, 0])
    with open(path) as f:
        for row in csv.DictReader(f):
            m = re.search(r"(k_\w+(<[^>]*>)?)", row["Kernel_Name"])
            k = (m.group(1) if m else row["Kernel_Name"][:40], row["Counter_Name"])
            acc[k][0] += float(row["Counter_Value"]); acc[k][1] += 1
    out.write("# %s\n" % path)
    for (kn, cn), (tot, n) in sorted(acc.items()):
        out.write("%-42s %-22s dispatches=%-4d mean=%.6g total=%.6g\n" % (kn, cn, n, tot / n, tot))
out.close()
