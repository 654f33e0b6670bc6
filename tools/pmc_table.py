#!/usr/bin/env python3
"""Merge the per-dispatch counter CSVs of tools/pmc_chain.sh: for every kernel NAME, the mean of each counter over its dispatches
with the same grid size (the chain repeats, so a (name, grid) pair is one layer's launch)."""
import csv, glob, os, re, sys
from collections import defaultdict, OrderedDict
root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
order = OrderedDict()
for f in sorted(glob.glob(os.path.join(root, "g*", "**", "*counter_collection.csv"), recursive=True)):
    for r in csv.DictReader(open(f)):
        name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0][:70]
        key = (name, int(r["Grid_Size"]) // max(int(r["Workgroup_Size"]), 1))
        order.setdefault(key, int(r.get("Dispatch_Id", 0)))
        acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = sorted({c for v in acc.values() for c in v})
print("kernel | workgroups | " + " | ".join(names))
for key in sorted(order, key=lambda k: order[k]):
    vals = []
    for c in names:
        v = acc[key].get(c)
        vals.append(f"{sum(v) / len(v):.4g}" if v else "-")
    print(f"{key[0]} | {key[1]} | " + " | ".join(vals))
