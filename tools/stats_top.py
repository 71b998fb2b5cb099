#!/usr/bin/env python3
"""Top kernels of a rocprofv3 --stats run.  usage: stats_top.py <dir> [divide_by] [n]"""
import csv
import glob
import os
import sys

d = sys.argv[1]
div = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
f = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot / 1e6 / div:.2f} ms per unit ({len(rows)} kernels)")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:top]:
    n = r["Name"].replace("(anonymous namespace)::", "")
    print("%9.3f ms %7d calls %10.1f us  %s" % (float(r["TotalDurationNs"]) / 1e6 / div, int(r["Calls"]), float(r["AverageNs"]) / 1e3, n[:80]))
