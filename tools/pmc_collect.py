#!/usr/bin/env python3
"""Average rocprofv3 counter values per launch for the kernels whose name contains <pattern>.
usage: pmc_collect.py <dir with pass*/ subdirs> <pattern>  -> JSON on stdout"""
import collections
import csv
import glob
import json
import os
import sys

d, pat = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: [0, 0.0])
dur = [0, 0.0]
for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        if pat in r["Kernel_Name"]:
            a = acc[r["Counter_Name"]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
for path in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        if pat in r["Kernel_Name"]:
            dur[0] += 1
            dur[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
res = {k: v[1] / v[0] for k, v in sorted(acc.items())}
if dur[0]:
    res["launch_us_profiled"] = dur[1] / dur[0]
json.dump(res, sys.stdout, indent=1)
print()
