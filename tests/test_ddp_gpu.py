"""GPU tier: the data-parallel step path with world_size 2. Both ranks share cuda:0 and exchange
gradients through gloo (RCCL needs one device per rank; the driver's 8-GPU run uses "nccl"): replicas
start identical (broadcast), run WGAN batches on different data with the two-phase generator backward
+ bucketed side-stream all-reduce, and must stay bit-identical; the averaged-gradient update must
equal a single-process emulation that accumulates both ranks' gradients and scales by 1/2."""
import os
import socket
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _build(dtype):
    import numpy as np
    import gan_inpainting_amd  # noqa: F401
    from gan_inpainting_amd.lib.models import networks
    from oracle import params as op
    G = networks.get_network("generator", "unet", dtype=dtype)
    G.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in op.make_unet_params(3).items()})
    D = networks.PatchGANDiscriminator(sigmoid=False, dtype=dtype)
    D.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in op.make_patchgan_params(4).items()})
    return G.to("cuda:0"), D.to("cuda:0")


def _batch(rank, it):
    import numpy as np
    from oracle import params as op
    g, m = op.synth_batch(1000 * rank + it, 2, 128, 128)
    return torch.from_numpy(g).cuda(), torch.from_numpy(m).cuda()


def _masks(it):
    g = torch.Generator().manual_seed(77 + it)
    return {5: (torch.rand((2, 512, 8, 8), generator=g) < 0.5).to(torch.uint8),
            6: (torch.rand((2, 512, 4, 4), generator=g) < 0.5).to(torch.uint8)}


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      GI_DIST_BACKEND="gloo")
    import gan_inpainting_amd  # noqa: F401
    from gan_inpainting_amd import optim, parallel, trainer
    parallel.init_from_env()
    G, D = _build("fp32")
    torch.distributed.broadcast(G.flat_params(), 0)
    torch.distributed.broadcast(D.flat_params(), 0)
    G.mark_dirty(), D.mark_dirty()
    sync = parallel.GradSync(bucket_floats=4 * 1024 * 1024)
    step = trainer.WGANStep(G, D, optim.RMSprop(G.parameters(), lr=5e-5), optim.RMSprop(D.parameters(), lr=5e-5), sync=sync,
                            overlap=True)   # critic on a side stream, as bench.py runs it
    for it in range(2):
        g, m = _batch(rank, it)
        G.impose_dropout_masks(_masks(it))
        step(g, m, it == 1)
    torch.cuda.synchronize()
    # numpy arrays travel by value (torch tensors would be passed as file descriptors of a process that exits)
    q.put((rank, G.flat_params().cpu().numpy(), D.flat_params().cpu().numpy()))
    torch.distributed.destroy_process_group()


def test_two_ranks_stay_identical_and_match_accumulated_gradients():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict((r, (torch.from_numpy(g), torch.from_numpy(d))) for r, g, d in [q.get(timeout=300) for _ in procs])
    for p in procs:
        p.join(timeout=60)
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1]), "replicas diverged"

    # single-process emulation: accumulate both ranks' gradients, update with scale 1/2
    import gan_inpainting_amd  # noqa: F401
    from gan_inpainting_amd import backend as B, optim, trainer
    G, D = _build("fp32")
    oG, oD = optim.RMSprop(G.parameters(), lr=5e-5, clamp=0.0), optim.RMSprop(D.parameters(), lr=5e-5, clamp=0.01)
    oG.grad_scale = oD.grad_scale = 0.5
    ops = trainer._Ops("cuda:0")
    for it in range(2):
        packs = []
        oD.zero_grad()
        for rank in range(2):            # critic step: per-rank forward with LOCAL BatchNorm statistics
            g, m = _batch(rank, it)
            mc, masked, inp = torch.empty_like(g), torch.empty_like(g), torch.empty_like(g)
            ops.mask_apply(g, m, mc, masked, True)
            G.impose_dropout_masks(_masks(it))
            gen, gs, gg = G._forward_raw(masked)
            ops.composite(masked, gen, mc, inp)
            dp = torch.empty((2, 1), device="cuda")
            L = torch.zeros(1, device="cuda")
            pr, s1, g1 = D._forward_raw(g)
            pf, s2, g2 = D._forward_raw(inp)
            ops.adv(pr, 2, 0.0, L, dp, +1.0)
            D._backward_raw(s1, g1, dp, False, True)
            ops.adv(pf, 2, 0.0, L, dp, -1.0)
            D._backward_raw(s2, g2, dp, False, True)
            packs.append((g, mc, inp, gs, gg))
        oD.step()
        if it == 1:
            oG.zero_grad()
            # NOTE: each rank's generator activations must still be live: 2 forwards <= n_slots (3)
            for (g, mc, inp, gs, gg) in packs:
                dp = torch.empty((2, 1), device="cuda")
                L = torch.zeros(1, device="cuda")
                p, s, gn = D._forward_raw(inp)
                ops.adv(p, 2, 0.0, L, dp, +1.0)
                d_adv = D._backward_raw(s, gn, dp, True, False)
                g_rec, tmp, g_gen = torch.empty_like(g), torch.empty_like(g), torch.empty_like(g)
                ops.recon("l1", inp, g, L, g_rec)
                ops.add(d_adv, g_rec, tmp)
                ops.mul(tmp, mc, g_gen)
                G._backward_raw(gs, gg, g_gen, False, True)
            oG.step()
    torch.cuda.synchronize()
    eg = (G.flat_params().cpu() - res[0][0]).abs().max().item()
    ed = (D.flat_params().cpu() - res[0][1]).abs().max().item()
    print("max |param diff| vs accumulated-gradient emulation: G", eg, "D", ed)
    # RMSprop's first step is lr*10*sign(g): elements whose gradient is at rounding level may flip
    assert ed <= 2.1e-3 and eg <= 2.1e-3
    frac_g = ((G.flat_params().cpu() - res[0][0]).abs() > 1e-6).float().mean().item()
    frac_d = ((D.flat_params().cpu() - res[0][1]).abs() > 1e-6).float().mean().item()
    if frac_g >= 2e-2:   # diagnostic: which tensors moved
        diff = (G.flat_params().cpu() - res[0][0]).abs()
        for t in G._inv:
            if t["kind"] <= 1:
                d = diff[t["offset"]: t["offset"] + t["numel"]]
                f = (d > 1e-6).float().mean().item()
                if f > 1e-3:
                    print(f"  {t['name']}: frac {f:.4f}")
    # With N=2 at 128x128 whole generator layers have gradients at rounding level (BatchNorm over 2-8 values),
    # and the fp32 atomics of the weight-gradient kernels order their sums differently from run to run: observed
    # fractions of sign-flipped elements range from 1e-5 to 8e-2 over repeated runs of the SAME code. A missing or
    # wrong all-reduce flips the sign of ~half of all elements (sign(g0) vs sign(g0+g1)); the bound separates the two.
    assert frac_g < 0.2 and frac_d < 2e-2, (frac_g, frac_d)
