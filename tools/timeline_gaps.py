#!/usr/bin/env python3
"""From a rocprofv3 --kernel-trace CSV: how much of the wall time between the first and the last kernel of the measured
part of a run the device had at least one kernel running, and the largest idle gaps with the kernels around them.
usage: timeline_gaps.py <dir with *_kernel_trace.csv> [skip_fraction]   (skip the first part of the trace: model set-up, warm-up)"""
import csv
import glob
import os
import sys

d = sys.argv[1]
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
rows = []
for path in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60], r.get("Queue_Id", r.get("Stream_Id", "?"))))
rows.sort()
t0, t1 = rows[0][0], max(r[1] for r in rows)
cut = t0 + (t1 - t0) * skip
rows = [r for r in rows if r[0] >= cut]
t0, t1 = rows[0][0], max(r[1] for r in rows)
busy, cur_end, gaps = 0, rows[0][0], []
prev = rows[0]
for s, e, n, q in rows:
    if s > cur_end:
        gaps.append((s - cur_end, prev[2], n))
        cur_end = s
    if e > cur_end:
        busy += e - cur_end
        cur_end = e
        prev = (s, e, n, q)
print(f"kernels {len(rows)}  wall {(t1 - t0) / 1e6:.1f} ms  busy {busy / 1e6:.1f} ms = {100.0 * busy / (t1 - t0):.1f} %  sum of kernel durations {sum(e - s for s, e, _, _ in rows) / 1e6:.1f} ms")
gaps.sort(reverse=True)
print("idle total %.2f ms in %d gaps; gaps > 50 us: %d (%.2f ms)" % (sum(g[0] for g in gaps) / 1e6, len(gaps), sum(1 for g in gaps if g[0] > 50000),
                                                                   sum(g[0] for g in gaps if g[0] > 50000) / 1e6))
for g, a, b in gaps[:12]:
    print(f"  {g / 1e3:8.1f} us  after {a[:44]:44s} before {b[:44]}")
import collections
byq = collections.defaultdict(lambda: [0, 0])
byn = collections.defaultdict(float)
for s_, e_, n_, q_ in rows:
    byq[q_][0] += 1
    byq[q_][1] += e_ - s_
    byn[n_] += e_ - s_
print("per queue:", {k: (v[0], round(v[1] / 1e6, 1)) for k, v in byq.items()})
for n_, v in sorted(byn.items(), key=lambda kv: -kv[1])[:8]:
    print(f"  {v / 1e6:9.1f} ms  {n_}")
# the five largest gaps: where in the window
big = sorted(gaps, reverse=True)[:5]
