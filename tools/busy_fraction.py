#!/usr/bin/env python3
"""GPU busy fraction of the benchmark's timed steps from a rocprofv3 kernel trace: union of kernel intervals / span, and the time two
kernels overlap (the critic runs on a side stream). usage: busy_fraction.py <kernel_trace.csv> <marker kernel substring> <steps>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
mark, steps = sys.argv[2], int(sys.argv[3])
idx = [i for i, r in enumerate(rows) if mark in r["Kernel_Name"]]
seg = rows[idx[-steps - 1]:idx[-1]]
t0, t1 = int(seg[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in seg)
ev = []
for r in seg:
    ev.append((int(r["Start_Timestamp"]), 1)); ev.append((int(r["End_Timestamp"]), -1))
ev.sort()
busy = over = 0; depth = 0; last = t0
for t, d in ev:
    if depth >= 1: busy += t - last
    if depth >= 2: over += t - last
    depth += d; last = t
tot = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg)
print(f"{steps} steps: span {(t1 - t0) / 1e3 / steps:.1f} us/step, busy {busy / 1e3 / steps:.1f} us/step ({100 * busy / (t1 - t0):.1f} %), "
      f"two kernels at once {over / 1e3 / steps:.1f} us/step, sum of kernel durations {tot / 1e3 / steps:.1f} us/step, kernels/step {len(seg) / steps:.1f}")
