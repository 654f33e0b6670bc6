#!/usr/bin/env python3
"""The generator's small-M layers (d5 d6 d7 u7 u6 u5 at 256x256, n = 32) through the single-layer C-ABI entries, HIP events, several
rounds: one line per layer with the kernel that served it. For A/B runs of igemm7 variants (ablation build: GI_LIB_PATH +
GI_IGEMM7_NSTG / GI_IGEMM7_PF / GI_IGEMM7_MAXSPLIT). usage: python tools/time_small.py [iters]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import layer_table as LT  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 50
rows = []
for k in (5, 6, 7):
    rows.append((f"d{k}",) + LT.conv(32, 256 >> k, 512, 512, iters))
for k, ca in ((7, 512), (6, 1024), (5, 1024)):
    rows.append((f"u{k}",) + LT.convT(32, 256 >> k, ca, 512, 0 if k == 7 else 1, iters))
rows.append(("u5dg",) + LT.conv(32, 8, 512, 1024, iters))
rows.append(("u6dg",) + LT.conv(32, 4, 512, 1024, iters))
rows.append(("d6dg",) + LT.convT(32, 4, 512, 512, 0, iters))
rows.append(("d5dg",) + LT.convT(32, 8, 512, 512, 0, iters))
print("  ".join(f"{n} {us:5.1f}us [{kern}]" for n, us, _, kern in rows), f"  sum {sum(r[1] for r in rows):.1f} us")
