#!/usr/bin/env python3
"""Turn gpurun_out/prof_<tag>/ (tools/profile_round.sh) into the committed summaries under profiles/."""
import csv, glob, json, os, re, sys


def newest(pattern):
    """gpurun merges every run into the same scratch directory: take the latest file."""
    return max(glob.glob(pattern), key=os.path.getmtime)


tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = f"gpurun_out/prof_{tag}"
os.makedirs("profiles", exist_ok=True)


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("_ZN12_GLOBAL__N_1", "")
    n = re.sub(r"void at::native::.*?(vectorized_elementwise_kernel|distribution_elementwise_grid_stride_kernel).*", r"torch \1 (host-side bookkeeping)", n)
    return n[:120]


def stats(path, title, out):
    rows = list(csv.DictReader(open(path)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    out.write(f"## {title}\n\nsource: `rocprofv3 --kernel-trace --stats` ({os.path.basename(path)}); total kernel time {tot/1e6:.2f} ms\n\n")
    out.write("| kernel | calls | total ms | avg us | min us | max us | % |\n|---|---|---|---|---|---|---|\n")
    for r in rows[:28]:
        out.write(f"| `{short(r['Name'])}` | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.3f} | {float(r['AverageNs'])/1e3:.1f} | "
                  f"{float(r['MinNs'])/1e3:.1f} | {float(r['MaxNs'])/1e3:.1f} | {float(r['Percentage']):.2f} |\n")
    out.write("\n")
    return rows


def pmc(path, counter):
    rows = [r for r in csv.DictReader(open(path)) if ("igemm5_kernel" in r["Kernel_Name"] or "igemm3_kernel" in r["Kernel_Name"]) and r["Counter_Name"] == counter]
    vals = [float(r["Counter_Value"]) for r in rows]
    return sum(vals) / len(vals), len(vals)


with open(f"profiles/{tag}_summary.md", "w") as out:
    out.write(f"# rocprofv3 summaries, round {tag} (MI355X, gfx950)\n\nCollected by `tools/profile_round.sh {tag}` through gpurun; "
              "raw CSVs are under `gpurun_out/` (scratch). Kernel names are the mangled template instantiations "
              "(`igemm_kernelIDF16_Li1E...` = `igemm_kernel<_Float16, PHASE=1, BM, BN, WGM, WGN>`).\n\n")
    ko = stats(newest(f"{src}/kernel_only/runc/*_kernel_stats.csv"),
               "A. `python bench.py --kernel-only --kernel-iters 50` — the dominant kernel alone (the roofline object's launch)", out)
    stats(newest(f"{src}/bench/runc/*_kernel_stats.csv"), "B. `python bench.py --steps 10 --warmup 5 --no-cpu-baseline` — whole benchmark process", out)
    fetch, nf = pmc(newest(f"{src}/pmc_fetch/runc/*_counter_collection.csv"), "FETCH_SIZE")
    write, nw = pmc(newest(f"{src}/pmc_write/runc/*_counter_collection.csv"), "WRITE_SIZE")
    dom = [r for r in ko if ("igemm5_kernel" in r["Name"] or "igemm3_kernel" in r["Name"])][0]
    avg_ms = float(dom["AverageNs"]) / 1e6
    flop = 2.0 * 4 * 32768 * 128 * 2048
    traffic = (2 * fetch + write) * 1024
    algo = 32 * 32 * 32 * 512 * 2 + 16 * 512 * 128 * 2 + 32 * 64 * 64 * 128 * 2
    out.write("## C. HBM traffic of the dominant kernel (separate `--pmc` passes)\n\n")
    out.write(f"* `rocprofv3 --pmc FETCH_SIZE`: mean {fetch:.0f} KB over {nf} launches; `--pmc WRITE_SIZE`: mean {write:.0f} KB over {nw} launches.\n")
    out.write("* gfx950 correction (guides/MI355X_MICROARCH.md, HBM): FETCH_SIZE counts 128-B requests at 64 B for wide coalesced reads → ×2; WRITE_SIZE exact.\n")
    out.write(f"* traffic per launch = (2·FETCH_SIZE + WRITE_SIZE)·1024 = **{traffic/1e6:.1f} MB**; algorithmic bytes (input 33.6 MB + weights 2.1 MB + output 33.6 MB) = {algo/1e6:.1f} MB "
              f"→ {traffic/algo:.2f}× (XCD-aware tile order: the phases / N tiles that share input rows run back to back on one XCD; DESIGN.md 4.1).\n")
    out.write(f"* rocprof average duration of the dominant kernel: **{avg_ms*1e3:.1f} us** → {flop/avg_ms/1e9:.0f} TFLOP/s = {flop/avg_ms/1e9/2500*100:.1f} % of 2.5 PFLOP/s dense fp16.\n")
json.dump({"hbm_bytes_per_launch": traffic, "fetch_size_kb": fetch, "write_size_kb": write, "correction": "2*FETCH_SIZE + WRITE_SIZE (gfx950)",
           "rocprof_avg_ms": avg_ms, "round": tag}, open("profiles/dominant_kernel_traffic.json", "w"), indent=1)
for name in ("kernel_only", "bench"):
    p = newest(f"{src}/{name}/runc/*_kernel_stats.csv")
    rows = list(csv.reader(open(p)))
    with open(f"profiles/{tag}_{name}_kernel_stats.csv", "w", newline="") as f:
        w = csv.writer(f)
        for r in rows:
            r[0] = short(r[0])
            w.writerow(r)
print(open(f"profiles/{tag}_summary.md").read()[:3000])
