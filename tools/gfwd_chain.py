#!/usr/bin/env python3
"""profiles/<tag>_generator_forward_chain.md from a `rocprofv3 --kernel-trace` of tools/gfwd_only.py.
usage: gfwd_chain.py <kernel_trace.csv> <tag>"""
import csv, os, re, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rows = list(csv.DictReader(open(sys.argv[1])))
tag = sys.argv[2]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "c1_gather" in r["Kernel_Name"]]
seg = rows[idx[-2] - 1:idx[-1] - 1]     # the saved-input copy precedes the first convolution
t0 = int(seg[0]["Start_Timestamp"])


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    m = re.match(r"_ZN12_GLOBAL__N_1\d+([a-z0-9_]+?)(I|E)", n)
    if m:
        n = m.group(1)
    return n.split("(")[0][:44]


def klass(k):
    if "bn_apply" in k: return "BatchNorm apply + activation (HBM passes)"
    if "bn_finalize" in k: return "BatchNorm finalize (one small launch per layer)"
    if "splitk" in k: return "split-K finish"
    if "igemm5" in k or "igemm3" in k: return "halo / LDS-DMA implicit GEMMs (d2-d4, u5-u2)"
    if "igemm_kernel" in k: return "generic implicit GEMM, split-K (d5-d7, u7-u6)"
    if "c1_" in k: return "single-channel layers d1 / u1"
    return "copies, dropout masks"


out = [f"# Generator forward, one call: kernel chain (round {tag})\n",
       "`rocprofv3 --kernel-trace -- python3 tools/gfwd_only.py 30` on one MI355X; U-Net 256x256, bs=32, fp16, train mode; the last",
       "forward of the run. Kernels run back to back on one stream; `us` is start-to-start, so it includes the dispatch gap.\n",
       "| # | start us | us | workgroups | kernel |", "|---|---|---|---|---|"]
tot, cls = 0.0, {}
for i, r in enumerate(seg):
    s = (int(r["Start_Timestamp"]) - t0) / 1000
    nxt = int(seg[i + 1]["Start_Timestamp"]) if i + 1 < len(seg) else int(r["End_Timestamp"])
    d = (nxt - int(r["Start_Timestamp"])) / 1000
    tot += d
    wg = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) // (int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"]))
    k = short(r["Kernel_Name"])
    out.append(f"| {i + 1} | {s:.1f} | {d:.1f} | {wg} x {r['Workgroup_Size_X']} | `{k}` |")
    cls[klass(k)] = cls.get(klass(k), 0.0) + d
out.append(f"\nTotal {tot:.1f} us over {len(seg)} kernels.\n")
out.append("| class | us | % |\n|---|---|---|")
for k, v in sorted(cls.items(), key=lambda kv: -kv[1]):
    out.append(f"| {k} | {v:.1f} | {100 * v / tot:.1f} |")
path = os.path.join(R, "profiles", f"{tag}_generator_forward_chain.md")
open(path, "w").write("\n".join(out) + "\n")
print("\n".join(out[-11:]))
