#!/usr/bin/env python3
"""Measured HBM copy ceiling of this GPU (torch tensor copy, 1 GiB read + 1 GiB write), for DESIGN.md context."""
import time
import torch
x = torch.empty(1 << 28, device="cuda", dtype=torch.float32).uniform_(-1, 1)
y = torch.empty_like(x)
for _ in range(5):
    y.copy_(x)
torch.cuda.synchronize()
for rep in (20, 200):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(rep):
        y.copy_(x)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / rep
    print(f"copy 1 GiB x{rep}: {ms:.3f} ms  ->  {2 * x.numel() * 4 / ms / 1e9:.2f} TB/s (read+write)")
