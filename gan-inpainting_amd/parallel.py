"""Data-parallel gradient exchange: one process per GPU, replicated G and D, per-rank minibatch,
local BatchNorm statistics (the single-GPU reference has no SyncBN), SUM all-reduce of each
network's flat fp32 gradient buffer over RCCL (torch.distributed backend "nccl" on ROCm), issued
on a side stream in buckets; the optimizer divides by world_size through its grad_scale.

The reference itself is single-device (train.py:44-46): there is no collective to mirror. The
exchange step here is the standard data-parallel one named by BASELINE.json's north_star.

Bucket order follows backward completion: for the generator the decoder half of the flat buffer
is complete after gi_net_backward_phase(phase=1), so its all-reduce overlaps the encoder half.
The same class works on CPU tensors with the gloo backend (tests/test_parallel_cpu.py).
"""
import torch
import torch.distributed as dist


class GradSync:
    def __init__(self, bucket_floats=8 * 1024 * 1024, group=None, use_side_stream=True):
        if not dist.is_initialized():
            raise RuntimeError("GradSync needs an initialised torch.distributed process group")
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.bucket_floats = int(bucket_floats)
        self.use_side_stream = use_side_stream
        self._stream = None
        self._pending = []

    # ---- generic (CPU or GPU) -------------------------------------------------------------------
    def buckets(self, begin, end):
        out, o = [], begin
        while o < end:
            n = min(self.bucket_floats, end - o)
            out.append((o, n))
            o += n
        return out

    def _comm_stream(self, device):
        if self._stream is None:
            self._stream = torch.cuda.Stream(device=device)
        return self._stream

    def launch(self, flat, begin=0, end=None):
        """Start the SUM all-reduce of flat[begin:end] (asynchronously on GPU tensors)."""
        end = flat.numel() if end is None else end
        if self.world == 1 or end <= begin:
            return
        if flat.is_cuda and self.use_side_stream:
            cur = torch.cuda.current_stream(flat.device)
            comm = self._comm_stream(flat.device)
            comm.wait_stream(cur)               # gradients of this range are complete on `cur`
            with torch.cuda.stream(comm):
                for o, n in self.buckets(begin, end):
                    self._pending.append(dist.all_reduce(flat[o:o + n], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        else:
            for o, n in self.buckets(begin, end):
                self._pending.append(dist.all_reduce(flat[o:o + n], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def wait(self, device=None):
        """Make the current stream (or the host, for CPU tensors) wait for every launched bucket."""
        for w in self._pending:
            w.wait()
        self._pending = []
        if self._stream is not None and device is not None:
            torch.cuda.current_stream(device).wait_stream(self._stream)

    # ---- network-level helpers --------------------------------------------------------------------
    def all_reduce(self, net):
        flat = net.flat_grads()
        self.launch(flat)
        self.wait(flat.device if flat.is_cuda else None)

    def grad_scale(self):
        return 1.0 / self.world

    def broadcast_parameters(self, nets, src=0):
        """Make every replica start from rank `src`'s weights (and BatchNorm running statistics)."""
        if self.world == 1:
            return
        for net in nets:
            flat = getattr(net, "flat_params", None)
            tensors = [flat()] if flat is not None else [p.data for p in net.parameters()]
            tensors += [b for b in getattr(net, "buffers", lambda: [])() if torch.is_tensor(b) and b.is_floating_point()]
            for t in tensors:
                dist.broadcast(t, src, group=self.group)
            if hasattr(net, "mark_dirty"):
                net.mark_dirty()


def local_device():
    """CUDA device index of this rank: LOCAL_RANK, folded onto the visible devices."""
    import os
    n = max(torch.cuda.device_count(), 1)
    return int(os.environ.get("LOCAL_RANK", "0")) % n


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* (torch.distributed.run)."""
    import os
    if dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world == 1:
        return 0, 1
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # GI_DIST_BACKEND=gloo rehearses the multi-rank path with several ranks on ONE GPU (tests, 1-GPU boxes)
    backend = backend or os.environ.get("GI_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
    if torch.cuda.is_available():
        torch.cuda.set_device(local_device())
    dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world
