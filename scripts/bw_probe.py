"""HBM bandwidth probe with plain torch ops (read-only sum, copy) -- the practical ceiling the solver kernels are compared with."""
import torch, time
dev = "cuda:0"
for mb in (64, 365, 1024, 4096):
    n = mb * 1024 * 1024 // 8
    x = torch.empty(n, dtype=torch.float64, device=dev).normal_()
    y = torch.empty_like(x)
    for name, fn, bytes_ in (("sum", lambda: x.sum(), n * 8), ("copy", lambda: y.copy_(x), 2 * n * 8), ("axpy", lambda: y.add_(x), 3 * n * 8)):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        K = 20
        for _ in range(K): fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / K
        print(f"{mb:5d} MiB {name:5s} {dt*1e6:9.1f} us  {bytes_/dt/1e12:6.2f} TB/s")
