#!/usr/bin/env python3
"""Compare two rocprofv3 kernel_stats CSVs per kernel (ms per step): tools/stats_diff.py old.csv new.csv [steps_old steps_new]"""
import csv, sys, re
def load(p, steps):
    d = {}
    for r in csv.DictReader(open(p)):
        n = re.sub(r"\(anonymous namespace\)::", "", r["Name"]).replace("void ", "").split("(")[0]
        c, t = d.get(n, (0, 0.0))
        d[n] = (c + int(r["Calls"]) / steps, t + float(r["TotalDurationNs"]) / 1e6 / steps)
    return d
so = float(sys.argv[3]) if len(sys.argv) > 3 else 7.0
sn = float(sys.argv[4]) if len(sys.argv) > 4 else 7.0
a, b = load(sys.argv[1], so), load(sys.argv[2], sn)
rows = []
for k in set(a) | set(b):
    ca, ta = a.get(k, (0, 0.0)); cb, tb = b.get(k, (0, 0.0))
    rows.append((tb - ta, k, ca, ta, cb, tb))
rows.sort()
print("%-90s %8s %8s   %8s %8s   %8s" % ("kernel", "calls", "ms", "calls", "ms", "delta ms"))
for d, k, ca, ta, cb, tb in rows:
    if abs(d) >= 0.02:
        print("%-90s %8.1f %8.3f   %8.1f %8.3f   %+8.3f" % (k[:90], ca, ta, cb, tb, d))
print("TOTAL %.3f -> %.3f ms/step; launches %.0f -> %.0f" % (sum(v[1] for v in a.values()), sum(v[1] for v in b.values()),
                                                          sum(v[0] for v in a.values()), sum(v[0] for v in b.values())))
