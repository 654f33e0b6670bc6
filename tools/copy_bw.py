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
