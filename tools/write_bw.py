#!/usr/bin/env python3
"""Achievable HBM write / copy bandwidth on this box (context for the roofline)."""
import torch, time
n = 1 << 30  # 4 GiB of float32
a = torch.empty(n, dtype=torch.float32, device="cuda"); b = torch.empty_like(a)
for name, fn, nbytes in (("fill (write only)", lambda: a.fill_(1.0), 4 * n),
                         ("copy (read+write)", lambda: b.copy_(a), 8 * n),
                         ("sum (read only)", lambda: a.sum(), 4 * n)):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    print("%-20s %.0f GB/s" % (name, nbytes * 10 / (e0.elapsed_time(e1) * 1e-3) / 1e9))
