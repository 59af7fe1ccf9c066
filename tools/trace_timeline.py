#!/usr/bin/env python
"""Print the kernel timeline (start offset, duration, name) of the LAST step of a rocprofv3 --kernel-trace CSV: which launches overlap,
where the gaps are.  usage: trace_timeline.py <kernel_trace.csv> <first kernel name substring of a step> [max rows]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
key = sys.argv[2]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if key in r["Kernel_Name"]]
# a step begins at a `key` kernel that follows a different kernel
begins = [i for j, i in enumerate(starts) if j == 0 or starts[j - 1] != i - 1]
lim = int(sys.argv[3]) if len(sys.argv) > 3 else 400
b = begins[-2] if len(begins) > 1 else begins[-1]
e = begins[-1] if len(begins) > 1 else len(rows)
t0 = int(rows[b]["Start_Timestamp"])
prev_end = t0
for r in rows[b:e][:lim]:
    s, f = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print("%9.1f %8.1f  gap %7.1f  q%-3s %s" % (s / 1e3, (f - s) / 1e3, (s - prev_end) / 1e3, r.get("Queue_Id", "?"), r["Kernel_Name"][:70]))
    prev_end = max(prev_end, f)
print("step total %.1f us" % ((int(rows[e - 1]["End_Timestamp"]) - t0) / 1e3))
