#!/usr/bin/env python
"""What the memory system gives a pure streaming kernel at the residual stream's working-set sizes -- the yardstick for the
split-residual GEMM epilogues (out-proj / FFN2 read hi + lo and write hi + lo: 8 bytes per element besides the A operand).
    python tools/stream_probe.py      -> GB/s (read + write) of dst.copy_(src) and of an in-place a += b, per size"""
import statistics

import torch

dev = "cuda"


def timed(fn, n=20):
    fn(); fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


for rows in (8640, 15360, 30720, 138752, 1 << 20):
    a = torch.randn(rows, 768, device=dev)                      # fp32 = the bytes of the (hi, lo) pair
    b = torch.randn(rows, 768, device=dev)
    mb = a.numel() * 4 / 1e6
    t_copy = statistics.median(timed(lambda: b.copy_(a)) for _ in range(3))
    t_rmw = statistics.median(timed(lambda: a.add_(1.0)) for _ in range(3))
    t_add = statistics.median(timed(lambda: a.add_(b)) for _ in range(3))
    print(f"rows {rows:8d} ({mb:7.1f} MB per plane pair): copy {2 * mb / t_copy / 1e3:7.0f} GB/s ({t_copy * 1e6:7.1f} us) | "
          f"in-place a += 1  {2 * mb / t_rmw / 1e3:7.0f} GB/s ({t_rmw * 1e6:7.1f} us) | a += b {3 * mb / t_add / 1e3:7.0f} GB/s ({t_add * 1e6:7.1f} us)")
