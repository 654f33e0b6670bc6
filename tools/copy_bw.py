import torch
for mb in (16, 33, 67, 134, 268):
    n = mb * 1024 * 1024 // 2
    a = torch.randn(n, device="cuda", dtype=torch.float16); b = torch.empty_like(a)
    for _ in range(5): b.copy_(a)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): b.copy_(a)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    print(f"copy {mb} MB -> {mb} MB: {us:6.1f} us  {2 * mb * 1.048576 / us * 1e3 / 1e3:5.2f} TB/s")

# the same through a ring of buffers larger than the 256 MiB Infinity Cache: what a pass gets whose input and output are not in it
for mb in (33, 67, 134):
    n = mb * 1024 * 1024 // 2
    k = max(2, 1536 // (2 * mb))
    src = [torch.randn(n, device="cuda", dtype=torch.float16) for _ in range(k)]
    dst = [torch.empty_like(src[0]) for _ in range(k)]
    for i in range(k): dst[i].copy_(src[i])
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for r in range(4):
        for i in range(k): dst[i].copy_(src[i])
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / (4 * k) * 1e3
    print(f"cold copy {mb} MB -> {mb} MB (ring of {k}): {us:6.1f} us  {2 * mb * 1.048576 / us:5.2f} TB/s")
