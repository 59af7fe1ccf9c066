#!/usr/bin/env python
"""Average duration (us) of the kernels whose name contains one of the given substrings, from a rocprofv3 *kernel_stats.csv.
usage: kstat.py stats.csv substr [substr ...]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
out = []
for key in sys.argv[2:]:
    for r in rows:
        if key in r["Name"]:
            out.append("%s %.1f us x%s" % (key, float(r["AverageNs"]) / 1e3, r["Calls"]))
print("; ".join(out))
