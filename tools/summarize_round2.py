#!/usr/bin/env python3
"""Turn gpurun_out/prof_<tag>/ (tools/profile_round2.sh) into the committed summaries under profiles/.
usage: python tools/summarize_round2.py <tag> <commit>"""
import csv, glob, json, os, re, subprocess, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
commit = sys.argv[2] if len(sys.argv) > 2 else subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
src = f"gpurun_out/prof_{tag}"
os.makedirs("profiles", exist_ok=True)
DOM = os.environ.get("GI_DOM_KERNEL", "igemm8_kernel<1, true")      # the roofline object's kernel (generator u3 with the fused input ReLU; round 2: igemm6_kernel<1, 128, true)


def newest(pattern):
    g = glob.glob(pattern)
    return max(g, key=os.path.getmtime) if g else None


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.match(r"_ZN12_GLOBAL__N_1\d+([a-z0-9_]+?)I(.*)", n)
    if m:
        n = m.group(1) + "<" + m.group(2)[:28] + ">"
    n = re.sub(r"at::native::.*?(vectorized_elementwise_kernel|distribution_elementwise_grid_stride_kernel).*", r"torch \1 (host-side bookkeeping)", n)
    return n.split("(")[0][:100]


def stats_table(path, title, out, top=30):
    rows = list(csv.DictReader(open(path)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    out.write(f"## {title}\n\nsource: `rocprofv3 --kernel-trace --stats`; total kernel time {tot / 1e6:.2f} ms\n\n")
    out.write("| kernel | calls | total ms | avg us | min us | max us | % |\n|---|---|---|---|---|---|---|\n")
    for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:top]:
        out.write(f"| `{short(r['Name'])}` | {r['Calls']} | {float(r['TotalDurationNs']) / 1e6:.3f} | {float(r['AverageNs']) / 1e3:.1f} | "
                  f"{float(r['MinNs']) / 1e3:.1f} | {float(r['MaxNs']) / 1e3:.1f} | {100 * float(r['TotalDurationNs']) / tot:.2f} |\n")
    out.write("\n")
    return rows


def copy_stats(path, dst):
    rows = list(csv.reader(open(path)))
    with open(dst, "w", newline="") as f:
        w = csv.writer(f)
        for r in rows:
            r[0] = short(r[0]) if r[0] != "Name" else r[0]
            w.writerow(r)


def pmc_mean(path, counter):
    rows = [r for r in csv.DictReader(open(path)) if DOM in r["Kernel_Name"] and r["Counter_Name"] == counter]
    vals = [float(r["Counter_Value"]) for r in rows]
    return (sum(vals) / len(vals), len(vals)) if vals else (None, 0)


def chain(path, marker, out, title, note):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
    starts = [i for k, i in enumerate(idx) if k == 0 or i - idx[k - 1] > 20]
    seg = rows[starts[-2]:starts[-1]] if len(starts) >= 2 else rows[starts[-1]:]
    t0 = int(seg[0]["Start_Timestamp"])
    out.write(f"## {title}\n\n{note}\n\n| # | start us | us | workgroups | kernel |\n|---|---|---|---|---|\n")
    tot, agg = 0.0, {}
    for i, r in enumerate(seg):
        s = (int(r["Start_Timestamp"]) - t0) / 1000
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000
        tot += d
        wg = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) // (int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"]))
        k = short(r["Kernel_Name"])
        out.write(f"| {i + 1} | {s:.1f} | {d:.1f} | {wg} x {r['Workgroup_Size_X']} | `{k}` |\n")
        base = k.split("<")[0]
        agg.setdefault(base, [0, 0.0])
        agg[base][0] += 1
        agg[base][1] += d
    out.write(f"\n{len(seg)} kernels, sum of durations {tot:.1f} us\n\n| kernel family | launches | us | % |\n|---|---|---|---|\n")
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:16]:
        out.write(f"| `{k}` | {v[0]} | {v[1]:.1f} | {100 * v[1] / tot:.1f} |\n")
    out.write("\n")
    return tot, len(seg)


def last_json(path):
    for line in reversed(open(path).read().strip().splitlines()):
        if line.startswith("{"):
            return json.loads(line)
    return None


with open(f"profiles/{tag}_summary.md", "w") as out:
    out.write(f"# rocprofv3 summaries, round {tag} (MI355X, gfx950), commit {commit}\n\nCollected by `tools/profile_round2.sh {tag}` in one gpurun call (one box), "
              f"summarised by `tools/summarize_round2.py`; raw CSVs live under `gpurun_out/` (scratch).\n\n")
    p = newest(f"{src}/kernel_only/*/*_kernel_stats.csv")
    ko = stats_table(p, "A. `python bench.py --kernel-only --kernel-iters 50`: the dominant kernel alone (the roofline object's launch)", out, 6)
    copy_stats(p, f"profiles/{tag}_kernel_only_kernel_stats.csv")
    p = newest(f"{src}/bench/*/*_kernel_stats.csv")
    stats_table(p, "B. `python bench.py --steps 20 --warmup 5 --preheat 50 --no-cpu-baseline`: the whole benchmark process (includes the generator-forward and "
                   "dominant-kernel timing loops bench.py runs after the timed steps)", out)
    copy_stats(p, f"profiles/{tag}_bench_kernel_stats.csv")
    dom = [r for r in ko if DOM in r["Name"]][0]
    avg_ms = float(dom["AverageNs"]) / 1e6
    flop = 2.0 * 4 * 32768 * 128 * 2048
    fetch, nf = pmc_mean(newest(f"{src}/pmc_fetch/*/*_counter_collection.csv"), "FETCH_SIZE")
    write, nw = pmc_mean(newest(f"{src}/pmc_write/*/*_counter_collection.csv"), "WRITE_SIZE")
    traffic = (2 * fetch + write) * 1024
    algo = 32 * 32 * 32 * 512 * 2 + 16 * 512 * 128 * 2 + 32 * 64 * 64 * 128 * 2
    out.write("## C. HBM traffic of the dominant kernel (separate `--pmc` passes)\n\n")
    out.write(f"* `rocprofv3 --pmc FETCH_SIZE`: mean {fetch:.0f} KB over {nf} launches; `--pmc WRITE_SIZE`: mean {write:.0f} KB over {nw} launches.\n")
    out.write("* gfx950 correction (guides/MI355X_MICROARCH.md, HBM section): FETCH_SIZE counts 128-byte requests at 64 B for wide coalesced reads -> x2; WRITE_SIZE exact.\n")
    out.write(f"* traffic per launch = (2 FETCH_SIZE + WRITE_SIZE) * 1024 = **{traffic / 1e6:.1f} MB**; algorithmic bytes (input 33.6 MB + weights 2.1 MB + output 33.6 MB) = "
              f"{algo / 1e6:.1f} MB -> {traffic / algo:.2f}x.\n")
    out.write(f"* rocprof average duration of the dominant kernel: **{avg_ms * 1e3:.1f} us** -> {flop / avg_ms / 1e9:.0f} TFLOP/s = {flop / avg_ms / 1e9 / 2500 * 100:.1f} % of 2.5 PFLOP/s dense fp16.\n\n")
    json.dump({"hbm_bytes_per_launch": traffic, "fetch_size_kb": fetch, "write_size_kb": write, "correction": "2*FETCH_SIZE + WRITE_SIZE (gfx950)",
               "rocprof_avg_ms": avg_ms, "round": tag, "kernel": DOM + "> (generator u3)", "commit": commit},
              open("profiles/dominant_kernel_traffic.json", "w"), indent=1)
    # D. counters
    counters = {}
    for g in sorted(glob.glob(f"{src}/pmc_g*/")):
        f = newest(g + "*/*_counter_collection.csv")
        if not f:
            continue
        acc = {}
        for r in csv.DictReader(open(f)):
            if DOM in r["Kernel_Name"]:
                acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        for k, v in acc.items():
            counters[k] = sum(v) / len(v)
    json.dump({"kernel": DOM + "> on the u3 shape (bench.py --kernel-only)", "round": tag, "commit": commit, "per_launch_avg": counters},
              open(f"profiles/{tag}_dominant_kernel_pmc.json", "w"), indent=1, sort_keys=True)
    if counters.get("SQ_VALU_MFMA_BUSY_CYCLES") and counters.get("SQ_BUSY_CYCLES"):
        busy = counters["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0     # 256 CUs x 4 SIMDs
        out.write("## D. Counters of the dominant kernel (`profiles/%s_dominant_kernel_pmc.json`)\n\n" % tag)
        out.write(f"* MFMA-busy cycles per SIMD = SQ_VALU_MFMA_BUSY_CYCLES / 1024 = {busy:.0f}; elapsed at the NOMINAL 2.4 GHz = {avg_ms * 1e-3 * 2.4e9:.0f} cycles -> "
                  f"{busy / (avg_ms * 1e-3 * 2.4e9):.2f} of the launch.\n")
        if counters.get("GRBM_GUI_ACTIVE"):   # measured clock: GRBM_GUI_ACTIVE is the sum over the 8 XCDs (guides/MI355X_MICROARCH.md, DVFS give-back)
            cyc = counters["GRBM_GUI_ACTIVE"] / 8.0
            out.write(f"* measured clock: GRBM_GUI_ACTIVE / 8 = {cyc:.0f} cycles per launch (the counter pass's own launches) -> MFMA-busy **{busy / cyc:.2f}** of the "
                      f"elapsed cycles (round 1, igemm5: 0.38 at a measured 1.96 GHz; round 2, igemm6: 0.45 at that clock); the quotient cycles / duration reads "
                      f"{cyc / (avg_ms * 1e-3) / 1e9:.2f} GHz against the un-profiled duration (high on launches this short, as the guide warns).\n")
        for k in ("SQ_INSTS_MFMA", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM", "SQ_WAIT_INST_LDS", "SQ_LDS_BANK_CONFLICT"):
            if k in counters:
                out.write(f"* {k}: {counters[k]:.0f}\n")
        out.write("\n")
    # F. secondary workloads
    out.write("## E. Secondary workloads (plain lines, then kernel statistics)\n\n")
    lines = {}
    for wl in ("headline", "wgan_gp_128", "dual_d_256", "config5_512", "vgg_512"):
        f = f"{src}/line_{wl}.log"
        if os.path.exists(f):
            j = last_json(f)
            if j:
                lines[wl] = j
                out.write(f"* `{wl}`: **{j['value']:.0f} {j['unit']}**, {j['ms_per_step']:.3f} ms/step ({j['config']['workload'][:110]})\n")
    out.write("\n")
    json.dump(lines, open(f"profiles/{tag}_bench_lines.json", "w"), indent=1)
    for wl in ("wgan_gp_128", "dual_d_256", "config5_512", "vgg_512"):
        p = newest(f"{src}/stats_{wl}/*/*_kernel_stats.csv")
        if p:
            stats_table(p, f"E.{wl}: `python bench.py --workload {wl} --steps 10 --warmup 5 --preheat 10 --no-cpu-baseline`", out, 16)
            copy_stats(p, f"profiles/{tag}_{wl}_kernel_stats.csv")

with open(f"profiles/{tag}_generator_forward_chain.md", "w") as out:
    out.write(f"# Generator forward, one call: kernel chain (round {tag}, commit {commit})\n\n")
    tot, n = chain(newest(f"{src}/gfwd/*/*_kernel_trace.csv"), "c1_gather", out, "U-Net 256x256, bs=32, fp16, train mode",
                   "`rocprofv3 --kernel-trace -- python3 tools/gfwd_only.py 20` on one MI355X; the last forward of the run. Kernels run back to back on one "
                   "stream; `us` is the kernel's own duration.")
with open(f"profiles/{tag}_step_chains.md", "w") as out:
    out.write(f"# One WGAN batch on ONE stream: kernel chains (round {tag}, commit {commit})\n\n`tools/step_chain.py`, 256x256 bs=32 fp16; the benchmark itself "
              "runs the critic on a side stream (overlap), these single-stream chains show every kernel in order.\n\n")
    chain(newest(f"{src}/step_critic/*/*_kernel_trace.csv"), "mask_apply", out, "Critic-only batch (4 of 5 batches)", "generator forward (no gradient) + critic forward / backward / RMSprop + clip")
    chain(newest(f"{src}/step_gen/*/*_kernel_trace.csv"), "mask_apply", out, "Batch with a generator update (every 5th)", "critic update, then generator forward / critic forward / both backwards / RMSprop")
print(open(f"profiles/{tag}_summary.md").read()[:2500])
