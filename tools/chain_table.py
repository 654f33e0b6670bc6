#!/usr/bin/env python3
"""Kernel chain of the LAST repetition in a rocprofv3 kernel trace: start, duration, workgroups, kernel.
usage: chain_table.py <kernel_trace.csv> <marker substring: first kernel of a repetition>"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
mark = sys.argv[2]
idx = [i for i, r in enumerate(rows) if mark in r["Kernel_Name"]]
# repetitions start where the marker appears after a gap of other kernels: use the last two well-separated ones
starts = [i for k, i in enumerate(idx) if k == 0 or i - idx[k - 1] > 20]
seg = rows[starts[-2]:starts[-1]] if len(starts) >= 2 else rows[starts[-1]:]
t0 = int(seg[0]["Start_Timestamp"])


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    m = re.match(r"_ZN12_GLOBAL__N_1\d+([a-z0-9_]+?)I(.*)", n)
    if m:
        n = m.group(1) + " " + m.group(2)[:24]
    return n.split("(")[0][:56]


tot = 0.0
for r in seg:
    s = (int(r["Start_Timestamp"]) - t0) / 1000
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000
    tot += d
    wg = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) // (int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"]))
    print(f"{s:9.1f} {d:7.1f} us  wg={wg:6d}x{r['Workgroup_Size_X']:>4}  {short(r['Kernel_Name'])}")
print(f"kernels {len(seg)}, sum {tot:.1f} us, span {(int(seg[-1]['End_Timestamp']) - t0) / 1000:.1f} us")
