#!/usr/bin/env python3
"""Development aid (GPU box): what this part sustains for pure copies, pure writes and the 1-read : 3-write
mix of the forward diffusion step (X -> time-major copy + 2 hop planes), measured with plain torch kernels."""
import torch

dev = torch.device("cuda", 0)
n = 1024 * 1024 * 1024 // 4           # 1 GiB per array: far beyond the 256 MB Infinity Cache
x = torch.randn(n, device=dev)
y = torch.empty_like(x)
out3 = torch.empty(3, n, device=dev)


def timed(fn, nbytes, label, iters=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print(f"{label:34s} {ms * 1e3:8.1f} us  {nbytes / ms / 1e9:7.2f} TB/s")


timed(lambda: y.copy_(x), 2 * 4 * n, "copy (1 read : 1 write)")
timed(lambda: y.fill_(1.0), 4 * n, "fill (write only)")
timed(lambda: out3.copy_(x.unsqueeze(0).expand(3, n)), 4 * 4 * n, "broadcast copy (1 read : 3 writes)")
timed(lambda: torch.add(x, 1.0, out=y), 2 * 4 * n, "add scalar (1 read : 1 write)")
timed(lambda: x.sum(), 4 * n, "sum (read only)")
