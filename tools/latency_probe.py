#!/usr/bin/env python3
"""Per-call latency of the host-buffer entry points (what a drop-in user of .apply(numpy chunk) sees); run on the GPU box."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pyaudiodsptools_amd as adsp  # noqa: E402


def bench(f, reps=2000, warm=200):
    for _ in range(warm):
        f()
    t0 = time.perf_counter()
    for _ in range(reps):
        f()
    return (time.perf_counter() - t0) / reps * 1e6


for n, c in [(4096, 1), (512, 1), (4096, 2), (4096, 64), (4096, 1024)]:
    adsp.config.initialize(44100, n)
    dev = adsp.CreateLowCutFilter(800, channels=c)
    x1 = np.random.default_rng(0).uniform(-1, 1, n).astype(np.float32)
    xb = np.random.default_rng(0).uniform(-1, 1, (c, n)).astype(np.float32)
    row = {}
    if c == 1:
        row["dev.apply"] = bench(lambda: dev.apply(x1))
    row["apply_batch"] = bench(lambda: dev.apply_batch(xb), reps=500 if c > 16 else 2000)
    d_in = torch.from_numpy(xb).cuda()
    d_out = torch.empty_like(d_in)

    def on_device():
        dev.engine.apply_device(d_in, d_out, 1)
        torch.cuda.synchronize()
    row["apply_device+sync"] = bench(on_device, reps=500 if c > 16 else 2000)
    samples = n * c
    print(f"N={n} C={c}: " + ", ".join(f"{k} {v:.1f} us ({samples / v:.0f} Msamples/s)" for k, v in row.items()), flush=True)
