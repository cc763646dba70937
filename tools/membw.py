"""HBM bandwidth calibration with library kernels: copy (read+write), fill (write), sum (read) over sizes."""
import torch
dev = torch.device('cuda:0')
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
for mb in (64, 256, 1024, 4096):
    n = mb * (1 << 20) // 4
    x = torch.empty(n, device=dev, dtype=torch.float32).normal_()
    y = torch.empty_like(x)
    t = timeit(lambda: y.copy_(x)); print(f'{mb:5d} MB copy  {2 * n * 4 / t / 1e12:6.2f} TB/s (r+w)')
    t = timeit(lambda: y.fill_(1.0)); print(f'{mb:5d} MB fill  {n * 4 / t / 1e12:6.2f} TB/s (w)')
    t = timeit(lambda: x.sum()); print(f'{mb:5d} MB sum   {n * 4 / t / 1e12:6.2f} TB/s (r)')
    xb = x.to(torch.bfloat16)
    t = timeit(lambda: torch.add(x, x, out=y)); print(f'{mb:5d} MB add   {2 * n * 4 / t / 1e12:6.2f} TB/s (r+w, same input twice)')
