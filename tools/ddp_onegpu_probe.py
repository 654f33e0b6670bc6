"""Probe (GPU box, 1 GPU): can two ranks share cuda:0 with a gloo process group and all_reduce CUDA
tensors? If so the whole data-parallel step path can be rehearsed on one GPU."""
import os, sys, torch, torch.distributed as dist
import torch.multiprocessing as mp

def worker(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    t = torch.full((1000,), float(rank + 1), device="cuda:0")
    try:
        w = dist.all_reduce(t, async_op=True); w.wait()
        torch.cuda.synchronize()
        print(f"rank {rank}: gloo all_reduce on cuda tensor -> {t[0].item()} (expect 3.0)")
    except Exception as e:
        print(f"rank {rank}: FAILED {type(e).__name__}: {e}")
    dist.destroy_process_group()

if __name__ == "__main__":
    mp.get_context("spawn")
    procs = [mp.get_context("spawn").Process(target=worker, args=(r, 2, 29533)) for r in range(2)]
    [p.start() for p in procs]; [p.join(120) for p in procs]
