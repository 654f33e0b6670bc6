#!/usr/bin/env python3
"""Per-layer table of the headline step's GEMM-shaped launches: every Conv2d / ConvTranspose2d forward, input-gradient and
weight-gradient shape of the generator (n = 32) and of the stacked critic (n = 64) through the single-layer C-ABI entries, timed
with HIP events on one box: kernel that served it (gi_debug_last_kernel), us, TFLOP/s and the fraction of the 2.5 PFLOP/s dense
fp16 peak - FLOP per launch from the shapes. Writes profiles/<tag>_layer_table.md.
usage: python tools/layer_table.py <tag> [iters]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import gan_inpainting_amd  # noqa: E402,F401
from gan_inpainting_amd import backend as B  # noqa: E402

F16 = B.GI_F16
PEAK = 2500.0


def timed(call, iters):
    for _ in range(3):
        call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        call()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3   # us


def conv(n, hs, cin, cout, iters):
    """Conv2d 4x4/s2 (cin -> cout) onto the hs x hs grid: forward of an encoder level / input gradient of a decoder level"""
    lib, ctx = B.lib(), B.get_ctx()
    x = (torch.rand((n, 2 * hs, 2 * hs, cin), device="cuda") - 0.3).half()
    w = (torch.rand((cout, 4, 4, cin), device="cuda") * 2 - 1) * 0.02
    packed = torch.empty(cout * 16 * cin, dtype=torch.float16, device="cuda")
    B.check(lib.gi_pack_weights(ctx, F16, B.ptr(w), cout, cin, B.ptr(packed), None))
    out = torch.empty((n, hs, hs, cout), dtype=torch.float16, device="cuda")
    ws = torch.empty(16 << 20, dtype=torch.float32, device="cuda")
    us = timed(lambda: B.check(lib.gi_conv_s2_forward(ctx, F16, B.ptr(x), B.ptr(packed), B.ptr(out), n, 2 * hs, 2 * hs, cin, cin, cout, cout, 0, 0,
                                                      B.ptr(ws), ws.numel() * 4)), iters)
    return us, 2.0 * n * hs * hs * cout * 16 * cin, B.last_kernel()


def convT(n, hs, ca, cb, relu, iters):
    """ConvTranspose2d 4x4/s2 (ca -> cb) from the hs x hs grid: forward of a decoder level / input gradient of an encoder level"""
    lib, ctx = B.lib(), B.get_ctx()
    x = (torch.rand((n, hs, hs, ca), device="cuda") - 0.3).half()
    w = (torch.rand((ca, 4, 4, cb), device="cuda") * 2 - 1) * 0.02
    phase = torch.empty(ca * 16 * cb, dtype=torch.float16, device="cuda")
    B.check(lib.gi_pack_weights(ctx, F16, B.ptr(w), ca, cb, None, B.ptr(phase)))
    out = torch.empty((n, 2 * hs, 2 * hs, cb), dtype=torch.float16, device="cuda")
    ws = torch.empty(16 << 20, dtype=torch.float32, device="cuda")
    us = timed(lambda: B.check(lib.gi_convT_s2_forward(ctx, F16, B.ptr(x), B.ptr(phase), B.ptr(out), n, hs, hs, ca, ca, cb, cb, relu, 0,
                                                       B.ptr(ws), ws.numel() * 4)), iters)
    return us, 2.0 * 4 * n * hs * hs * cb * 4 * ca, B.last_kernel()


def wgrad(n, hs, ca, cb, relu_s, ldm, iters):
    lib, ctx = B.lib(), B.get_ctx()
    S = (torch.rand((n, hs, hs, ca), device="cuda") - 0.5).half()
    L = (torch.rand((n, 2 * hs, 2 * hs, ldm * cb), device="cuda") - 0.5).half()
    dW = torch.zeros((ca, 4, 4, cb), dtype=torch.float32, device="cuda")
    nbytes = lib.gi_wgrad_s2_scratch_bytes(F16, n, hs, hs, ca, cb)
    ws = torch.empty(max(nbytes // 4, 4), dtype=torch.float32, device="cuda")
    us = timed(lambda: B.check(lib.gi_wgrad_s2_ws(ctx, F16, B.ptr(S), B.ptr(L), B.ptr(dW), n, hs, hs, ca, ca, cb, ldm * cb, relu_s, 1.0, B.ptr(ws), nbytes)), iters)
    return us, 2.0 * n * hs * hs * ca * 16 * cb, B.last_kernel()


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    ch = [0, 64, 128, 256, 512, 512, 512, 512]
    rows = []
    for k in range(2, 8):   # generator encoder forward d2..d7: Conv2d(ch[k-1] -> ch[k]) onto the (256 >> k) grid
        rows.append((f"G d{k} fwd", "generator forward") + conv(32, 256 >> k, ch[k - 1], ch[k], iters))
    for k in range(7, 1, -1):   # decoder forward u7..u2: ConvTranspose2d(ca -> ch[k-1]) from the (256 >> k) grid
        ca = ch[7] if k == 7 else 2 * ch[k]
        rows.append((f"G u{k} fwd", "generator forward") + convT(32, 256 >> k, ca, ch[k - 1], 0 if k == 7 else 1, iters))
    for k in range(2, 8):   # decoder input gradients: Conv2d gather (ch[k-1] -> ca) onto the (256 >> k) grid
        ca = ch[7] if k == 7 else 2 * ch[k]
        rows.append((f"G u{k} dgrad", "generator backward") + conv(32, 256 >> k, ch[k - 1], ca, iters))
    for k in range(7, 1, -1):   # encoder input gradients: sub-pixel phases (ch[k] -> ch[k-1]) from the (256 >> k) grid
        rows.append((f"G d{k} dgrad", "generator backward") + convT(32, 256 >> k, ch[k], ch[k - 1], 0, iters))
    for k in range(2, 8):
        ca = ch[7] if k == 7 else 2 * ch[k]
        rows.append((f"G u{k} wgrad", "generator backward") + wgrad(32, 256 >> k, ca, ch[k - 1], 0 if k == 7 else 1, 1, iters))
        rows.append((f"G d{k} wgrad", "generator backward") + wgrad(32, 256 >> k, ch[k], ch[k - 1], 0, 2, iters))
    dch = [1, 64, 128, 256, 512]
    for i in range(2, 5):   # the stacked critic, n = 64
        rows.append((f"D conv{i} fwd", "critic") + conv(64, 256 >> i, dch[i - 1], dch[i], iters))
        rows.append((f"D conv{i} dgrad", "critic") + convT(64, 256 >> i, dch[i], dch[i - 1], 0, iters))
        rows.append((f"D conv{i} wgrad", "critic") + wgrad(64, 256 >> i, dch[i], dch[i - 1], 0, 1, iters))
    commit = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True, cwd=ROOT).stdout.strip() or os.environ.get("GI_COMMIT", "?")
    out = [f"# Per-layer table of the headline step's GEMM-shaped launches (round {tag}, commit {commit})\n",
           "`python tools/layer_table.py` on one MI355X: single-layer C-ABI entries (no fused epilogue operands), HIP events over "
           f"{iters} back-to-back launches per shape; FLOP = 2 x MAC of the layer (SURVEY.md 8a2); peak = 2.5 PFLOP/s dense fp16.\n",
           "| layer | part | kernel | us | TFLOP/s | fraction of peak |", "|---|---|---|---|---|---|"]
    for name, part, us, flop, kern in rows:
        tf = flop / us / 1e6
        out.append(f"| {name} | {part} | `{kern}` | {us:.1f} | {tf:.0f} | {tf / PEAK:.3f} |")
    agg = {}
    for name, part, us, flop, kern in rows:
        a = agg.setdefault(kern, [0, 0.0, 0.0])
        a[0] += 1; a[1] += us; a[2] += flop   # noqa: E702
    out += ["", "## By kernel instantiation (sum over the shapes above, largest time first)", "",
            "| kernel | shapes | us | TFLOP/s | fraction of peak |", "|---|---|---|---|---|"]
    for kern, (cnt, us, flop) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        tf = flop / us / 1e6
        out.append(f"| `{kern}` | {cnt} | {us:.1f} | {tf:.0f} | {tf / PEAK:.3f} |")
    fwd = [r for r in rows if r[1] == "generator forward"]
    out += ["", f"Generator forward GEMMs: {sum(r[2] for r in fwd):.1f} us for {sum(r[3] for r in fwd) / 1e9:.1f} GFLOP = "
                f"{sum(r[3] for r in fwd) / sum(r[2] for r in fwd) / 1e6:.0f} TFLOP/s."]
    for d in ("profiles", "gpurun_out"):      # (gpurun merges only gpurun_out/ back: copy that one into profiles/ afterwards)
        os.makedirs(os.path.join(ROOT, d), exist_ok=True)
        open(os.path.join(ROOT, d, f"{tag}_layer_table.md"), "w").write("\n".join(out) + "\n")
    print("\n".join(out))


if __name__ == "__main__":
    main()
