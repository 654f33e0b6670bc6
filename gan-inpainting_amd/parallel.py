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
    """bucket_floats=None (default): ONE collective per launched range - the ranges are the backward's completion phases
    (generator: decoder 26.5 M floats, inner encoder 12.6 M, outer encoder 2.7 M; critic: conv4 + head, conv3..1), large
    messages for 7 x 153 GB/s of xGMI per GPU; a number splits a range into buckets of that many floats.
    Every network has its own communication stream and pending list (`key`, default id(flat)), so a network's reduction is
    ordered only behind ITS gradients. The collectives themselves share ONE communicator (torch: the process group's RCCL
    communicator and its internal stream; abi: one ncclComm per GradSync): RCCL runs the operations of a communicator one
    after the other, in the order every rank issued them - the critic's and the generator's reductions do queue behind each
    other there, and all ranks must launch them in the same order (they do: the step schedule is identical on every rank).
    comm='torch' uses torch.distributed (RCCL behind backend "nccl", gloo on CPU); comm='abi' drives RCCL through the
    library's own C entry points (gi_comm_* / gi_allreduce_sum_f32: include/ganinpaint.h; executed on one rank only so
    far - experimental until a multi-GPU run has exercised it); its communicator is created by prepare(device), which the
    step classes call before the first batch, not in the middle of a step."""

    def __init__(self, bucket_floats=None, group=None, use_side_stream=True, comm=None, wire=None, wire_scale=1.0):
        if not dist.is_initialized():
            raise RuntimeError("GradSync needs an initialised torch.distributed process group")
        import os
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.bucket_floats = None if not bucket_floats else int(bucket_floats)
        self.use_side_stream = use_side_stream
        self.comm = comm or os.environ.get("GI_COMM", "torch")
        # wire='fp16' (83.6 MB instead of 167.3 MB per generator update): a launched range travels as
        # fp16(wire_scale * g) through a staging buffer and is written back as fp32 / wire_scale at wait(); the sum over the ranks
        # is then taken in fp16 (~1e-3 relative). Off by default (fp32 on the wire) until a scaling curve says the exchange is
        # exposed; wire_scale is the caller's guard against fp16 underflow of small gradients (the network's loss scale is a
        # natural choice: its fp16 backward already ran at that magnitude). comm='torch' only.
        self.wire = wire or os.environ.get("GI_WIRE", "fp32")
        if self.wire not in ("fp32", "fp16"):
            raise ValueError(f"GradSync: wire must be 'fp32' or 'fp16', not {self.wire!r}")
        if self.wire == "fp16" and self.comm == "abi":
            raise ValueError("GradSync: the fp16 wire format needs comm='torch'")
        self.wire_scale = float(wire_scale)
        self._streams, self._pending, self._events = {}, {}, {}
        self._stage = {}
        self._abi = None

    # ---- generic (CPU or GPU) -------------------------------------------------------------------
    def buckets(self, begin, end):
        if self.bucket_floats is None:
            return [(begin, end - begin)] if end > begin else []
        out, o = [], begin
        while o < end:
            n = min(self.bucket_floats, end - o)
            out.append((o, n))
            o += n
        return out

    def _comm_stream(self, device, key):
        if key not in self._streams:
            self._streams[key] = torch.cuda.Stream(device=device)
        return self._streams[key]

    def _abi_comm(self, device):
        """Lazily created RCCL communicator behind the C-ABI: rank 0 draws the unique id, torch.distributed carries it."""
        if self._abi is None:
            from . import backend as B
            import ctypes as C
            lib = B.lib()
            uid = torch.zeros(128, dtype=torch.uint8)
            if self.rank == 0:
                buf = (C.c_char * 128)()
                B.check(lib.gi_comm_unique_id(buf))
                uid = torch.frombuffer(bytearray(buf.raw), dtype=torch.uint8).clone()
            holder = uid.to(device) if dist.get_backend(self.group) == "nccl" else uid
            dist.broadcast(holder, 0, group=self.group)
            raw = bytes(holder.cpu().numpy().tobytes())
            h = C.c_void_p()
            B.check(lib.gi_comm_create(raw, self.rank, self.world, device.index or 0, C.byref(h)))
            self._abi = h
        return self._abi

    def prepare(self, device):
        """Create what the first launch() would otherwise create lazily in the middle of a step (the C-ABI communicator: a
        collective rendezvous of all ranks)."""
        if self.comm == "abi" and self.world > 1 and torch.device(device).type == "cuda":
            self._abi_comm(torch.device(device))

    def _launch_fp16(self, flat, o, n, key, pend):
        """One range in the fp16 wire format: stage = fp16(scale * g); the reduced stage is written back by wait()."""
        stage = self._stage.get((key, o, n))
        if stage is None:
            stage = self._stage[(key, o, n)] = torch.empty(n, dtype=torch.float16, device=flat.device)
        stage.copy_(flat[o:o + n] * self.wire_scale)
        work = dist.all_reduce(stage, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        pend.append(_Fp16Range(work, flat, o, n, stage, 1.0 / self.wire_scale))

    def launch(self, flat, begin=0, end=None, key=None):
        """Start the SUM all-reduce of flat[begin:end] (asynchronously on GPU tensors)."""
        end = flat.numel() if end is None else end
        if self.world == 1 or end <= begin:
            return
        key = id(flat) if key is None else key
        pend = self._pending.setdefault(key, [])
        if self.wire == "fp16":
            if flat.is_cuda and self.use_side_stream:
                cur = torch.cuda.current_stream(flat.device)
                comm = self._comm_stream(flat.device, key)
                comm.wait_stream(cur)
                with torch.cuda.stream(comm):
                    for o, n in self.buckets(begin, end):
                        self._launch_fp16(flat, o, n, key, pend)
            else:
                for o, n in self.buckets(begin, end):
                    self._launch_fp16(flat, o, n, key, pend)
            return
        if flat.is_cuda and self.use_side_stream:
            cur = torch.cuda.current_stream(flat.device)
            comm = self._comm_stream(flat.device, key)
            comm.wait_stream(cur)               # gradients of this range are complete on `cur`
            with torch.cuda.stream(comm):
                for o, n in self.buckets(begin, end):
                    if self.comm == "abi":
                        from . import backend as B
                        B.check(B.lib().gi_allreduce_sum_f32(self._abi_comm(flat.device), flat.data_ptr() + 4 * o, n, comm.cuda_stream))
                    else:
                        pend.append(dist.all_reduce(flat[o:o + n], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        else:
            for o, n in self.buckets(begin, end):
                pend.append(dist.all_reduce(flat[o:o + n], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def wait(self, device=None, key=None, flat=None):
        """Make the current stream (or the host, for CPU tensors) wait for the launched ranges of one network (key /
        flat) or, with neither, of every network."""
        if flat is not None and key is None:
            key = id(flat)
        keys = list(self._pending) if key is None else [key]
        back = []
        for k in keys:
            for w in self._pending.get(k, []):
                w.wait()
                if isinstance(w, _Fp16Range):
                    back.append(w)
            self._pending[k] = []
        if device is not None:
            for k in (list(self._streams) if key is None else [key]):
                if k in self._streams:
                    torch.cuda.current_stream(device).wait_stream(self._streams[k])
        for w in back:          # fp16 wire: the reduced stage back into the fp32 gradient range (on the waiting stream)
            w.write_back()

    # ---- network-level helpers --------------------------------------------------------------------
    def all_reduce(self, net):
        flat = net.flat_grads()
        self.launch(flat)
        self.wait(flat.device if flat.is_cuda else None, flat=flat)

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

    def close(self):
        if self._abi is not None:
            from . import backend as B
            B.lib().gi_comm_destroy(self._abi)
            self._abi = None


class _Fp16Range:
    """A launched range in the fp16 wire format: the collective's work handle + where its result goes."""

    def __init__(self, work, flat, o, n, stage, inv_scale):
        self.work, self.flat, self.o, self.n, self.stage, self.inv_scale = work, flat, o, n, stage, inv_scale

    def wait(self):
        self.work.wait()

    def write_back(self):
        dst = self.flat[self.o:self.o + self.n]
        dst.copy_(self.stage)
        if self.inv_scale != 1.0:
            dst.mul_(self.inv_scale)


def _rendezvous_watchdog(rank, seconds):
    """The process group's rendezvous either completes or the rank says so and exits: a daemon thread that, unless cancelled,
    prints which rank is stuck after `seconds` and ends THIS process with code 5 (nothing is re-executed; a launcher - bench.py's
    launch_ranks or torch.distributed.run - then sees a failed rank instead of a silent hang). Returns the cancel function."""
    import os
    import sys
    import threading
    done = threading.Event()

    def watch():
        if not done.wait(seconds):
            print(f"gan_inpainting_amd.parallel: rank {rank} is still in the process-group rendezvous after {seconds:.0f} s "
                  f"(MASTER_ADDR={os.environ.get('MASTER_ADDR')} MASTER_PORT={os.environ.get('MASTER_PORT')} "
                  f"WORLD_SIZE={os.environ.get('WORLD_SIZE')}); giving up", file=sys.stderr, flush=True)
            os._exit(5)
    threading.Thread(target=watch, daemon=True).start()
    return done.set


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
    # GI_RENDEZVOUS_TIMEOUT (seconds, default 120): a rank that cannot join says which one it is and exits non-zero
    cancel = _rendezvous_watchdog(rank, float(os.environ.get("GI_RENDEZVOUS_TIMEOUT", "120")))
    try:
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    finally:
        cancel()
    status = os.environ.get("GI_RANK_STATUS_DIR")      # bench.py's launch_ranks: which ranks came up
    if status:
        with open(os.path.join(status, f"rank{rank}.ready"), "w") as f:
            f.write(str(os.getpid()))
    return rank, world
