#!/usr/bin/env python3
"""One-rank RCCL communicator through the C-ABI (gi_comm_*): prints a line per stage so that a hang can be located.
Run it under `timeout`: RCCL's bootstrap needs a usable network interface (NCCL_SOCKET_IFNAME=lo on a box without one)."""
import ctypes as C
import os
import sys
import time

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import torch  # noqa: E402

import gan_inpainting_amd  # noqa: E402,F401
from gan_inpainting_amd import backend as B  # noqa: E402

t0 = time.time()


def say(msg):
    print(f"[{time.time() - t0:6.2f}s] {msg}", flush=True)


lib = B.lib()
torch.zeros(1, device="cuda")
say("cuda up")
uid = (C.c_char * 128)()
B.check(lib.gi_comm_unique_id(uid))
say("unique id drawn")
comm = C.c_void_p()
B.check(lib.gi_comm_create(bytes(uid.raw), 0, 1, 0, C.byref(comm)))
say("communicator created")
side = torch.cuda.Stream()
buf = torch.arange(1 << 20, dtype=torch.float32, device="cuda")
ref = buf.clone()
side.wait_stream(torch.cuda.current_stream())
B.check(lib.gi_allreduce_sum_f32(comm, B.ptr(buf), buf.numel(), side.cuda_stream))
say("all-reduce issued")
B.check(lib.gi_allreduce_wait(comm, side.cuda_stream, torch.cuda.current_stream().cuda_stream))
torch.cuda.synchronize()
say("all-reduce done, identity on one rank: %s" % bool(torch.equal(buf, ref)))
B.check(lib.gi_comm_destroy(comm))
say("destroyed")
print("RCCL_ONE_RANK_OK" if torch.equal(buf, ref) else "RCCL_ONE_RANK_WRONG", flush=True)

# ---- stage 2: ranges of a bound network's gradient buffer, then a second communicator through GradSync(comm='abi') ----
from gan_inpainting_amd.lib.models import networks  # noqa: E402
D = networks.PatchGANDiscriminator(sigmoid=False, image_size=64, dtype="fp32").cuda()
D(torch.rand(2, 1, 64, 64, device="cuda")).sum().backward()
g0 = D.flat_grads().clone()
say("net ready")
uid = (C.c_char * 128)()
B.check(lib.gi_comm_unique_id(uid))
comm = C.c_void_p()
B.check(lib.gi_comm_create(bytes(uid.raw), 0, 1, 0, C.byref(comm)))
say("second communicator created")
split = lib.gi_net_phase_split(D._handle)
side.wait_stream(torch.cuda.current_stream())
B.check(lib.gi_net_allreduce_grads_async(D._handle, comm, split, -1, side.cuda_stream))
B.check(lib.gi_net_allreduce_grads_async(D._handle, comm, 0, split, side.cuda_stream))
B.check(lib.gi_allreduce_wait(comm, side.cuda_stream, torch.cuda.current_stream().cuda_stream))
torch.cuda.synchronize()
say("net ranges reduced: %s" % bool(torch.equal(D.flat_grads(), g0)))
B.check(lib.gi_comm_destroy(comm))
import socket  # noqa: E402
import torch.distributed as dist  # noqa: E402
from gan_inpainting_amd import parallel  # noqa: E402
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
say("gloo group up")
sync = parallel.GradSync(comm="abi")
h = sync._abi_comm(torch.device("cuda", 0))
sync.world = 2     # force the exchange path on the one-rank communicator
say("GradSync communicator created")
sync.launch(D.flat_grads())
say("GradSync launched")
sync.wait(torch.device("cuda", 0), flat=D.flat_grads())
torch.cuda.synchronize()
say("GradSync done: %s" % bool(torch.equal(D.flat_grads(), g0)))
sync.close()
dist.destroy_process_group()
print("STAGE2_OK", flush=True)
