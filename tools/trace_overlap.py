#!/usr/bin/env python3
"""Reads a rocprofv3 kernel trace (CSV) and reports how much the kernels overlap in time:
    python tools/trace_overlap.py <dir with *_kernel_trace.csv>
Prints the busy time (union of kernel intervals), the sum of kernel durations and, per kernel name, the share
of its duration during which a kernel of another stream (queue) was running as well."""
import csv
import sys
from pathlib import Path

rows = []
for f in Path(sys.argv[1]).rglob("*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-40:], r.get("Queue_Id", "")))
rows.sort()
t0 = rows[0][0]
union, cur_s, cur_e = 0, None, None
for s, e, _, _ in rows:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            union += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
union += cur_e - cur_s
total = sum(e - s for s, e, _, _ in rows)
print(f"kernels {len(rows)}  span {(rows[-1][1] - t0) / 1e6:.2f} ms  busy (union) {union / 1e6:.2f} ms  sum of durations {total / 1e6:.2f} ms")
per = {}
for i, (s, e, n, q) in enumerate(rows):
    ov = 0
    for s2, e2, n2, q2 in rows:
        if q2 == q or e2 <= s or s2 >= e:
            continue
        ov += min(e, e2) - max(s, s2)
    d = per.setdefault(n, [0, 0, 0])
    d[0] += 1
    d[1] += e - s
    d[2] += min(ov, e - s)
for n, (c, dur, ov) in sorted(per.items(), key=lambda kv: -kv[1][1]):
    print(f"  {n:40s} calls {c:4d}  mean {dur / c / 1e6:8.3f} ms  overlapped by another queue {100.0 * ov / max(dur, 1):5.1f} %")
if len(rows) < 80:
    for s, e, n, q in rows:
        print(f"    {(s - t0) / 1e6:9.3f} .. {(e - t0) / 1e6:9.3f} ms  q{q}  {n}")
